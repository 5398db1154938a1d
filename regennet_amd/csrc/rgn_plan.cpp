// Planning and dispatch (host): which kernels run one denoiser evaluation (plan_eval) and in which arithmetic a sampling loop runs its steps
// (prec_plan: plain bf16 | plain fp16 | split-bf16), the enqueueing of an evaluation (run_layers / run_eval: kernel chains on the handle's streams,
// capturable), and the sampling loop itself (sample_range: one launch per phase where k_layers<true> applies, hipGraph replays otherwise).
// Reference: diffusion/gaussian_diffusion.py:610-742, 891-1005 (the loops), model/cmdm.py:173-252 (the evaluation).
#include "rgn_host.h"

namespace rgnh {

Dims make_dims(const rgn_ctx* c, int B, bool guided) {
    Dims dm;
    dm.B = B;
    dm.Bm = guided ? 2 * B : B;
    dm.T = c->cfg.num_frames;
    dm.Tq = c->Tq;
    dm.etd = c->etd;
    dm.F = c->F;
    dm.d = c->d;
    dm.H = c->H;
    dm.dh = c->d / c->H;
    dm.ff = c->ff;
    dm.L = c->L;
    return dm;
}

inline int default_tail(int S, int layers, bool etd = false) {
    if (etd) return S;   // emb_trans_dec feeds the timestep/condition embedding in as a token: measured 10x more sensitive
                         // to bulk-phase rounding under guidance (2 layers, 50-step DDIM + CFG: 1.9e-3 with 40 of 50 steps split)
    // What the bulk phase may cost is empirical (tests/test_hip_parity.py sweeps the switch point against the reference, and
    // DESIGN.md §6 tabulates models of other depths): the last step returns the denoiser's own prediction (coef1[0] = 1,
    // coef2[0] = 0), so earlier rounding reaches the result only through the network's sensitivity to x_t, which the
    // LayerNorm stack damps - the deeper the model the more. Measured with 10 (of 1000 DDPM) / 8 (of 100 DDIM + CFG) split
    // steps: 8 layers 6.7e-5 / 1.1e-4, 4 layers 1.6e-4, 2 layers 5.2e-4 / 8.3e-4; a 20-step DDIM schedule with 8 of them
    // split measured 3e-3 on a tiny 2-layer model (which the depth scaling now keeps split-bf16 throughout), while the 8-layer
    // 20-step goldens measure 1.2e-4 with 5 and 1.0e-4 with all 20 steps split. The 8-layer curves are flat from 5 split steps on
    // (1.2e-4 / 1.2e-4 / 1.0e-4 / 1.2e-4 with 5 on the four sweeps vs 0.7 - 1.2e-4 with 10, 1.4 - 3.7e-4 with 2), the shallow
    // models' are not (2 layers: 5 -> 1.2e-3 / 1.4e-3). So: max(5, S / 200) split-bf16 steps for models of >= 8 layers,
    // max(8, S / 100) * 8 / layers for shallower ones.
    // Short schedules - the reference's shipped evaluation setting is 5 steps (`--timestep_respacing ddim5` through p_sample_loop,
    // README.md:134-137) - measured on the reference's own 5-step outputs (tests: test_reference_evaluation_setting_switch_point_sweep,
    // three 8-layer goldens, both kernel forms): 4.3 - 5.0e-5 with all 5 steps split, 4.6 - 5.6e-5 with 3, 6.6 - 7.9e-5 with 2, ~1e-3
    // with 1, 2e-2 with none. Up to 10 steps: 3 split-bf16 steps (the other steps then reach the plain-bf16 kernels: 10.7 -> 7.7 ms per
    // 5-step call at B = 256).
    if (layers >= 8) {
        if (S <= 10) return S < 3 ? S : 3;
        const int t8 = (S + 199) / 200 < 5 ? 5 : (S + 199) / 200;
        return t8 < S ? t8 : S;
    }
    int t = (S + 99) / 100;
    t = t < 8 ? 8 : t;
    t = (t * 8 + layers - 1) / (layers > 0 ? layers : 1);
    return t < S ? t : S;
}

GemmArgs gemm_args(const rgn_ctx* c, const Lin& L, const float* A, int lda, float* C, int ldc, int M) {
    GemmArgs g{};
    g.A = A;
    g.lda = lda;
    g.W = c->dp<float>(L.w);
    g.Whi = c->dp<uint16_t>(L.hi);
    g.Wlo = c->dp<uint16_t>(L.lo);
    g.bias = L.has_bias ? c->dp<float>(L.b) : nullptr;
    g.add = nullptr;
    g.ldadd = 0;
    g.add_mod = 0;
    g.C = C;
    g.ldc = ldc;
    g.M = M;
    g.N = L.N;
    g.K = L.K;
    g.Kp = L.Kp;
    g.act = 0;
    return g;
}

// x [B,F,T] -> token-major GEMM operand: fp32 xin (F32 mode) or split K32-blocked planes (both guidance halves)
int pack_state(rgn_ctx* c, const float* x, const Dims& dm, bool guided, hipStream_t s) {
    if (c->cfg.precision == RGN_PREC_F32) {
        RGN_LAUNCH(c, KC_UPDATE, s, launch_pack_x(x, c->xin, Planes{nullptr, nullptr, 0}, 1, dm, s));
    } else {
        const Planes xp{c->xin_hi, has_lo(c) ? c->xin_lo : nullptr, dm.Bm * dm.Tq};
        if (xp.rows != c->xin_rows) {   // the blocked layout depends on the row count: K-padding columns must read as zero
            const size_t bytes = (size_t)2 * c->cfg.max_batch * c->Tq * align_up((size_t)c->F, 32) * 2;
            RGN_HIP(c, hipMemsetAsync(c->xin_hi, 0, bytes, s));
            RGN_HIP(c, hipMemsetAsync(c->xin_lo, 0, bytes, s));
            c->xin_rows = xp.rows;
        }
        RGN_LAUNCH(c, KC_UPDATE, s, launch_pack_x(x, nullptr, xp, guided ? 2 : 1, dm, s));
    }
    return RGN_OK;
}

// ---- the plan of one denoiser evaluation: WHICH kernels run it. The one place that decides - run_layers / run_eval /
//      rgn_sample_range dispatch on it and rgn_plan_query reports it (launches, algorithmic FLOPs and L2 weight-stream bytes per
//      kernel class), so that what bench.py prices is by construction what the engine launches.
enum AttnForm { AF_LAYERS = 0, AF_QKV, AF_QKV_LONG, AF_ROWGEMM_ATTN, AF_GEMM_ATTN, AF_PLAIN };
enum TailForm { TF_LAYERS = 0, TF_MLP_X3, TF_MLP, TF_ROWGEMM, TF_GEMM_LN };
struct EvalPlan {
    bool sb = false;          // small-batch engine (k_sb_gemm chain) for the whole evaluation
    bool layers = false;      // k_layers: the whole decoder stack in one kernel, one sample per workgroup
    bool steps = false;       // k_layers<true>: whole runs of sampler steps in one launch (sampling only)
    bool step_fused = false;  // k_step: output projection + sampler update + next input embedding (sampling only)
    AttnForm attn = AF_PLAIN;
    TailForm tail = TF_GEMM_LN;
};
bool all_frag(const rgn_ctx* c) {
    bool ok = true;
    for (int l = 0; l < c->L; ++l) ok = ok && c->layers[l].qkv.fr && c->layers[l].out.fr && c->layers[l].ff1.fr && c->layers[l].ff2.fr;
    return ok;
}
inline bool eval_x3_phase(const rgn_ctx* c, bool phase_x3) {
    return c->cfg.precision == RGN_PREC_BF16X3 || (c->cfg.precision == RGN_PREC_BF16_X3TAIL && phase_x3);
}
// Can this evaluation end in the fused step boundary (rgn_step.hip)? Sampling step of the plain-bf16 phase on the
// throughput kernels with hi-only residual planes; guided and unguided (guided sampling runs one k_step over the conditional rows
// after the chains have joined).
bool step_fusable(const rgn_ctx* c, bool x3, int rows) {
    return c->step_fused && !x3 && c->cfg.precision == RGN_PREC_BF16_X3TAIL && !use_sb(c, rows) && !c->bulk_resid_lo;
}
// dm: the WHOLE evaluation (Bm = all rows of all chains); x3: split-bf16 arithmetic for it; sampling: inside a sampler loop
EvalPlan plan_eval(const rgn_ctx* c, const Dims& dm, bool guided, bool x3, bool sampling) {
    EvalPlan p;
    const int prec = c->cfg.precision, rows = dm.Bm * dm.Tq;
    const bool fast = prec != RGN_PREC_F32;
    const bool lo_planes = prec == RGN_PREC_BF16X3 || prec == RGN_PREC_BF16_X3TAIL;     // has_lo()
    const bool h_lo = x3 || (lo_planes && c->bulk_resid_lo);                              // residual stream carries a lo plane
    p.sb = use_sb(c, rows);
    if (p.sb) {
        p.attn = (c->sb_attn && sb_qkv_attn_supported(c->d, dm.dh, dm.Tq)) ? AF_QKV : AF_GEMM_ATTN;
        return p;
    }
    p.step_fused = sampling && step_fusable(c, x3, rows);
    p.layers = fast && !x3 && c->layers_fused && dm.Bm >= c->layers_min_b && !h_lo && all_frag(c);
    const bool motion_per_wg = c->layers_guided == 2 || (c->layers_guided == 1 && dm.Bm > c->num_cus);   // (rgn_host.h: where a motion per workgroup pays)
    p.steps = p.layers && p.step_fused && c->layers_steps && (!guided || (motion_per_wg && c->ffn_hi &&
              // the guided form parks a motion's conditional x0 (6 x 4096 floats) in the idle hidden-tensor planes: 2 max_batch Tq ffp bf16
              (size_t)2 * c->cfg.max_batch * c->Tq * align_up((size_t)c->ff, 32) * 2 >= (size_t)dm.B * 6 * 4096 * 4));
    if (p.layers) {
        p.attn = AF_LAYERS;
        p.tail = TF_LAYERS;
        return p;
    }
    const bool fr0 = c->L > 0 && c->layers[0].qkv.fr != 0;
    if (fast && c->fuse_qkv) p.attn = AF_QKV;
    else if (fast && c->qkv_long && !x3 && fr0 && (size_t)rows * c->layers[0].qkv.Kp * 2 < (1ull << 31)) p.attn = AF_QKV_LONG;
    else if (fast && c->attn_x3 && !x3 && c->rowgemm && dm.dh % 32 == 0 && fr0) p.attn = AF_ROWGEMM_ATTN;
    else if (fast && c->attn_x3) p.attn = AF_GEMM_ATTN;
    else p.attn = AF_PLAIN;
    const bool frlo = c->L > 0 && c->layers[0].out.fr_lo && c->layers[0].ff1.fr_lo && c->layers[0].ff2.fr_lo;
    if (fast && x3 && c->mlp_x3 && lo_planes && frlo) p.tail = TF_MLP_X3;
    else if (fast && !x3 && c->mlp && !h_lo) p.tail = TF_MLP;
    else if (fast && !x3 && c->rowgemm) p.tail = TF_ROWGEMM;
    else p.tail = TF_GEMM_LN;
    return p;
}

// ---- the precision plan of a sampling loop over the bound schedule: loop indices [0, tail) run split-bf16, [tail, tail + n16) plain
//      fp16 operands, the rest plain bf16. Why three phases: v_mfma_f32_32x32x16_f16 has the bf16 instruction's nominal rate and 8x less operand
//      rounding, but the chip is power-managed under a matrix load and a pure f16 MFMA loop sustains 7.5 - 8 % less than the bf16 one
//      (tools/experiments/mfma_sustained.hip: 1690 vs 1830 TFLOP/s) - k_layers measures -6 % end to end on fp16 operands. The sampler contracts
//      what early steps get wrong, so fp16 is spent where rounding still reaches the output: the last plain steps. With them on fp16
//      the split-bf16 tail, at 3.7x the cost of a plain step, shrinks from 5 (3 for schedules of <= 10 steps) to F16_TAIL steps at the same
//      error on every golden (tools/f16_sweep.py, tests: test_three_phase_precision_schedule_sweep).
constexpr int F16_STEPS_DEFAULT = 8, F16_TAIL = 2;
PrecPlan prec_plan(const rgn_ctx* c, const Dims& dm, bool guided) {
    PrecPlan pp;
    if (c->cfg.precision != RGN_PREC_BF16_X3TAIL) return pp;
    const EvalPlan plain = plan_eval(c, dm, guided, false, true);
    // (the forms with an fp16 instantiation: the one-kernel stack - multi-step, or per evaluation in front of k_step - and the kernel-per-stage chain of 150-frame models -
    //  k_qkv_attn_long + k_mlp2 + k_step - and the same chain at <= 64 tokens below the one-kernel stack's batch threshold: k_qkv_attn_rs + k_mlp2 +
    //  k_step - whose planes hand the residual stream from step to step)
    const bool f16_ok = c->bulk_f16 && ((plain.layers && plain.step_fused) || (plain.step_fused && !plain.layers && plain.tail == TF_MLP &&
                                                       (plain.attn == AF_QKV_LONG || (plain.attn == AF_QKV && c->qkv_rs && c->d == 512 && c->L > 0 && c->layers[0].qkv.fr16))));
    pp.n16 = !f16_ok ? 0 : (c->f16_steps >= 0 ? c->f16_steps : F16_STEPS_DEFAULT);
    if (c->x3_tail >= 0) pp.tail = c->x3_tail;
    else if (pp.n16 > 0 && c->L >= 8 && !c->etd) pp.tail = F16_TAIL < c->S ? F16_TAIL : c->S;
    else pp.tail = default_tail(c->S, c->L, c->etd != 0);
    if (pp.tail > c->S) pp.tail = c->S;
    if (pp.n16 > c->S - pp.tail) pp.n16 = c->S - pp.tail;
    return pp;
}

int run_layers_sb(rgn_ctx* c, const Dims& dm, bool sampling, const float* cond_rows, const float* ccond_rows, hipStream_t s) {
    const int d = c->d, Ld = c->L * c->d, M = dm.Bm * dm.Tq;
    const bool x3 = eval_x3(c);
    auto base = [&](const Lin& L) {
        SbArgs g{};
        g.Whi = c->dp<__bf16>(L.hi); g.Wlo = c->dp<__bf16>(L.lo); g.w_rows = L.N;
        g.bias = L.has_bias ? c->dp<float>(L.b) : nullptr;
        g.M = M; g.N = L.N; g.Kp = L.Kp; g.Tq = dm.Tq;
        return g;
    };
    {   // input embedding + hoisted condition part: tmp = xin . Wx'^T + c0
        SbArgs g = base(c->lin_x);
        g.Ahi = c->xin_hi; g.Alo = c->xin_lo; g.a_rows = M;
        g.resid = c->c0; g.ldr = d; g.C = c->tmp; g.ldc = d;
        RGN_LAUNCH(c, KC_SB, s, launch_sb_gemm(g, 0, 0, x3, s));
    }
    if (c->etd) {
        const Planes none{nullptr, nullptr, 0};
        if (sampling)
            RGN_LAUNCH(c, KC_EMBED, s, launch_emb_rows(cond_rows, c->te_all, c->d_step, c->dp<float>(c->off_pe), c->tmp, none, dm, c->cfg.wo_pos_emb, s));
        else
            RGN_LAUNCH(c, KC_EMBED, s, launch_emb_rows(c->emb, nullptr, nullptr, c->dp<float>(c->off_pe), c->tmp, none, dm, c->cfg.wo_pos_emb, s));
    }
    const Planes att_p{c->att_hi, x3 ? c->att_lo : nullptr, M};
    const bool fused_attn = plan_eval(c, dm, false, x3, sampling).attn == AF_QKV;
    for (int l = 0; l < c->L; ++l) {
        const LayerW& w = c->layers[l];
        {   // layer input = norm3 of the previous layer (layer 0: the embedding itself); in_proj -> q (pre-scaled), k, v
            SbArgs g = base(w.qkv);
            g.src = c->tmp; g.xout = c->h;
            if (l) { g.ga = c->dp<float>(c->layers[l - 1].ln[4]); g.ba = c->dp<float>(c->layers[l - 1].ln[5]); }
            g.Qhi = c->q_hi; g.Khi = c->k_hi; g.Vhi = c->vt_hi;
            if (x3) { g.Qlo = c->q_lo; g.Klo = c->k_lo; g.Vlo = c->vt_lo; }
            g.d = d; g.H = c->H; g.dh = dm.dh; g.Tqp = c->Tqp; g.qscale = 1.0f / sqrtf((float)dm.dh);
            if (fused_attn) {   // ... and the attention, a (sample, head) per workgroup: one launch, q / k / v stay in LDS
                g.att = att_p;
                RGN_LAUNCH(c, KC_QKV, s, launch_sb_qkv_attn(g, dm.Bm, x3, s));
            } else {
                RGN_LAUNCH(c, KC_SB, s, launch_sb_gemm(g, 1, 2, x3, s));
            }
        }
        if (!fused_attn) {
            AttnX3Args a{};
            a.Qhi = c->q_hi; a.Qlo = c->q_lo; a.Khi = c->k_hi; a.Klo = c->k_lo; a.Vthi = c->vt_hi; a.Vtlo = c->vt_lo;
            a.out = att_p;
            a.Bm = dm.Bm; a.H = c->H; a.dh = dm.dh; a.d = d; a.Tq = dm.Tq; a.Tqp = c->Tqp; a.x3 = x3;
            RGN_LAUNCH(c, KC_ATTN, s, launch_attn_x3(a, s));
        }
        {   // tmp = attention . Wo^T + bo + h
            SbArgs g = base(w.out);
            g.Ahi = c->att_hi; g.Alo = c->att_lo; g.a_rows = M;
            g.resid = c->h; g.ldr = d; g.C = c->tmp; g.ldc = d;
            RGN_LAUNCH(c, KC_SB, s, launch_sb_gemm(g, 0, 0, x3, s));
        }
        {   // h = norm2(norm1(tmp) + folded cross-attention); ffn = gelu(h . W1^T + b1)
            SbArgs g = base(w.ff1);
            g.src = c->tmp; g.xout = c->h;
            g.ga = c->dp<float>(w.ln[0]); g.ba = c->dp<float>(w.ln[1]); g.gb = c->dp<float>(w.ln[2]); g.bb = c->dp<float>(w.ln[3]);
            g.pervec = sampling ? (ccond_rows ? ccond_rows + (size_t)l * d : nullptr) : c->call + (size_t)l * d;
            g.ldper = Ld;
            g.stepvec = sampling ? c->call_time + (size_t)l * d : nullptr;
            g.ldstep = Ld; g.d_step = c->d_step;
            g.Chi = c->ffn_hi; g.Clo = x3 ? c->ffn_lo : nullptr; g.c_rows = M;
            RGN_LAUNCH(c, KC_SB, s, launch_sb_gemm(g, 1, 1, x3, s));
        }
        {   // tmp = ffn . W2^T + b2 + h
            SbArgs g = base(w.ff2);
            g.Ahi = c->ffn_hi; g.Alo = c->ffn_lo; g.a_rows = M;
            g.resid = c->h; g.ldr = d; g.C = c->tmp; g.ldc = d;
            RGN_LAUNCH(c, KC_SB, s, launch_sb_gemm(g, 0, 0, x3, s));
        }
    }
    SbArgs g = base(c->lin_out);   // x0tok = norm3(tmp) . Wout^T + bout
    g.src = c->tmp;
    g.ga = c->dp<float>(c->layers[c->L - 1].ln[4]); g.ba = c->dp<float>(c->layers[c->L - 1].ln[5]);
    g.C = c->x0tok; g.ldc = c->F;
    RGN_LAUNCH(c, KC_SB, s, launch_sb_gemm(g, 1, 0, x3, s));
    return RGN_OK;
}

// Embedding GEMM + the L decoder layers + output projection for samples [s0, s0+ns) of the evaluation's sample list
// (row range [s0*Tq, (s0+ns)*Tq)), enqueued on stream s. F32 mode is always called with the full range.
// Arguments of k_layers that do not depend on the launch's sample range but for the per-sample vector base (rgn_layers.hip)
void fill_layers_args(rgn_ctx* c, LayersArgs& g, const Dims& dm, bool sampling, const float* ccond_rows, int s0, bool f16 = false) {
    const int Ld = c->L * c->d;
    g.Tq = dm.Tq; g.L = c->L;
    for (int l = 0; l < c->L; ++l) {
        const LayerW& w = c->layers[l];
        LayerWts& t = g.lw[l];
        t.Wqkv = c->dp<__bf16>(f16 ? w.qkv.fr16 : w.qkv.fr); t.Wo = c->dp<__bf16>(f16 ? w.out.fr16 : w.out.fr);
        t.W1 = c->dp<__bf16>(f16 ? w.ff1.fr16 : w.ff1.fr); t.W2 = c->dp<__bf16>(f16 ? w.ff2.fr16 : w.ff2.fr);
        t.bqkv = c->dp<float>(w.qkv.b); t.bo = c->dp<float>(w.out.b); t.bf1 = c->dp<float>(w.ff1.b); t.bf2 = c->dp<float>(w.ff2.b);
        t.g1 = c->dp<float>(w.ln[0]); t.b1 = c->dp<float>(w.ln[1]); t.g2 = c->dp<float>(w.ln[2]); t.b2 = c->dp<float>(w.ln[3]);
        t.g3 = c->dp<float>(w.ln[4]); t.b3 = c->dp<float>(w.ln[5]);
    }
    g.pervec = sampling ? (ccond_rows ? ccond_rows + (size_t)s0 * Ld : nullptr) : c->call + (size_t)s0 * Ld;
    g.ldper = Ld;
    g.stepvec = sampling ? c->call_time : nullptr;
    g.ldstep = Ld; g.d_step = c->d_step;
    g.qscale = 1.0f / sqrtf((float)dm.dh);
}

int run_layers(rgn_ctx* c, const Dims& dmf, bool guided, bool sampling, const float* cond_rows, const float* ccond_rows,
               int s0, int ns, hipStream_t s) {
    const int prec = c->cfg.precision;
    const int d = c->d, Ld = c->L * c->d, Mtot = dmf.Bm * dmf.Tq, Mb = dmf.B * dmf.Tq;
    const int row0 = s0 * dmf.Tq, M = ns * dmf.Tq;
    const bool fast = prec != RGN_PREC_F32, x3 = eval_x3(c);
    const EvalPlan pl = plan_eval(c, dmf, guided, x3, sampling);
    if (pl.sb) return run_layers_sb(c, dmf, sampling, cond_rows, ccond_rows, s);   // (called with the full range)
    const bool f16 = !x3 && sampling && c->phase_f16;      // the schedule's fp16 sub-phase (rgn_sample_range sets it only where prec_plan allows)
    Dims dm = dmf;
    dm.Bm = ns;
    // ---- the big GEMMs: F32 mode keeps fp32 activations (k_gemm_f32); the bf16 modes chain pre-split
    //      K32-blocked planes between kernels (k_gemm_x3, DMA-fed). Plane pointers are advanced by row0 rows
    //      (32 elements each) while Planes::rows stays the row count of the whole evaluation.
    // x3: this evaluation's GEMMs form three MFMAs per product and read hi + lo planes. In the bulk phase of the precision
    // schedule every plane is written hi-only (the residual stream's lo plane is an option, bulk_resid_lo).
    auto pln = [&](__bf16* hi, __bf16* lo, bool with_lo) {
        return Planes{fast ? hi + (size_t)row0 * 32 : nullptr, (fast && with_lo) ? lo + (size_t)row0 * 32 : nullptr, Mtot};
    };
    const Planes none{nullptr, nullptr, 0};
    const bool h_lo = x3 || (has_lo(c) && c->bulk_resid_lo);
    const Planes xin_p = pln(c->xin_hi, c->xin_lo, has_lo(c)), h_p = pln(c->h_hi, c->h_lo, h_lo), att_p = pln(c->att_hi, c->att_lo, x3),
                 ffn_p = pln(c->ffn_hi, c->ffn_lo, x3);
    float* h = c->h + (size_t)row0 * d;
    float* tmp = c->tmp + (size_t)row0 * d;
    float* qkv = c->qkv + (size_t)row0 * 3 * d;
    float* att = c->att + (size_t)row0 * d;
    float* ffn = c->ffn + (size_t)row0 * c->ff;
    // The residual stream: fp32 `h` in F32 mode (and for the fused-LN variant, whose kernel reads it), added in the GEMM
    // epilogue. In the bf16 modes only its split planes exist: k_layernorm adds hi + lo to the GEMM output it normalises
    // and writes planes only, so neither kernel touches an fp32 copy (31 MB less HBM traffic per LayerNorm at B=256).
    const bool h32 = !fast;
    auto big = [&](const Lin& L, const float* A32, int lda, const Planes& Ap, float* C, int ldc, const Planes& Cp,
                   const float* add, int act, int rows) -> int {
        if (!fast) {
            GemmArgs g = gemm_args(c, L, A32, lda, C, ldc, rows);
            g.add = add;
            g.ldadd = d;
            g.act = act;
            RGN_LAUNCH(c, KC_GEMM, s, launch_gemm(g, small_prec(c), s));
        } else {
            GemmX3Args g{};
            g.Ahi = Ap.hi; g.Alo = Ap.lo; g.a_rows = Ap.rows;
            g.Whi = c->dp<__bf16>(L.hi); g.Wlo = c->dp<__bf16>(L.lo);
            g.bias = L.has_bias ? c->dp<float>(L.b) : nullptr;
            g.add = add; g.ldadd = d;
            g.C = C; g.ldc = ldc;
            g.Chi = Cp.hi; g.Clo = Cp.lo; g.c_rows = Cp.rows;
            g.M = rows; g.N = L.N; g.Kp = L.Kp; g.act = act;
            // tile choice: 256x256 (one workgroup per CU, ~1.45x faster loop) only when the chain's launch has enough
            // tiles to take the CUs through more than one round, so that epilogues overlap the next round's loops:
            // measured -13 % at 3840 rows per chain (B=256), +2 % at 7680 (B=512, CFG at B=256), +5 % at 15360 (B=1024)
            const int variant = (rows >= c->big_tile_rows && L.N % 256 == 0) ? 1 : 0;
            RGN_LAUNCH(c, KC_GEMM, s, launch_gemm_x3(g, x3, variant, s));
        }
        return RGN_OK;
    };
    int rc;
    // input embedding + hoisted condition part (InputProcess/fuse/pos-enc, cmdm.py:201-218)
    if (c->skip_embed_out) {
        // fused step boundary (k_step): the residual-stream planes already hold this evaluation's input embedding
    } else if (fast) {   // xin planes and c0 already hold both guidance halves
        if ((rc = big(c->lin_x, nullptr, 0, xin_p, h32 ? h : nullptr, d, h_p, c->c0 + (size_t)row0 * d, 0, M))) return rc;
    } else {
        if ((rc = big(c->lin_x, c->xin, c->F, none, c->h, d, none, c->c0, 0, Mb))) return rc;
        if (guided)
            RGN_HIP(c, hipMemcpyAsync(c->h + (size_t)Mb * d, c->h, (size_t)Mb * d * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    if (c->etd) {
        if (sampling)
            RGN_LAUNCH(c, KC_EMBED, s, launch_emb_rows(cond_rows ? cond_rows + (size_t)s0 * d : nullptr, c->te_all, c->d_step,
                                                      c->dp<float>(c->off_pe), h, h_p, dm, c->cfg.wo_pos_emb, s));
        else
            RGN_LAUNCH(c, KC_EMBED, s, launch_emb_rows(c->emb + (size_t)s0 * d, nullptr, nullptr, c->dp<float>(c->off_pe), h, h_p, dm,
                                                      c->cfg.wo_pos_emb, s));
    }
    const size_t slab0 = (size_t)s0 * c->H * c->Tqp * dm.dh;      // attention-ready planes: first slab of this range
    bool layers_done = false;
    if (pl.layers) {
        // plain-bf16 phase, <= 64 tokens, d = 512 / ff = 1024 / 4 heads: ALL layers in one kernel, one sample per workgroup - the residual
        // stream stays in LDS from the input embedding to the last norm3, only the weights stream (rgn_layers.hip)
        LayersArgs g{};
        g.h = h_p.hi; g.out = h_p.hi; g.rows = h_p.rows; g.Bm = ns;
        fill_layers_args(c, g, dm, sampling, ccond_rows, s0, f16);
        g.f16 = f16 ? 1 : 0;
        RGN_LAUNCH(c, KC_LAYERS, s, launch_layers(g, s));
        layers_done = true;
    }
    for (int l = layers_done ? c->L : 0; l < c->L; ++l) {
        const LayerW& w = c->layers[l];
        if (pl.attn == AF_QKV) {
            // in_proj + attention in one kernel (two samples x half the heads per workgroup): q, k, v only ever exist in LDS
            QkvAttnArgs g{};
            g.Ahi = h_p.hi; g.Alo = h_p.lo; g.a_rows = h_p.rows;
            g.Whi = c->dp<__bf16>(w.qkv.hi); g.Wlo = c->dp<__bf16>(w.qkv.lo);
            g.Wfr = (w.qkv.fr && c->qkv_rs) ? c->dp<__bf16>(f16 ? w.qkv.fr16 : w.qkv.fr) : nullptr;   // plain phase: weights streamed to registers
            g.Wfr_lo = (x3 && w.qkv.fr && w.qkv.fr_lo && c->qkv_rs && !c->qkv_x3_dma) ? c->dp<__bf16>(w.qkv.fr_lo) : nullptr;   // ... and the split phase's (k_qkv_attn_rs_x3)
            g.f16 = f16 ? 1 : 0;
            g.bias = c->dp<float>(w.qkv.b);
            g.out = att_p;
            g.Bm = ns; g.Kp = w.qkv.Kp; g.d = d; g.H = c->H; g.Tq = dm.Tq;
            g.qscale = 1.0f / sqrtf((float)dm.dh);
            g.Bm_eval = dmf.Bm;   // samples of the WHOLE evaluation (all kernel chains), not of this chain
            RGN_LAUNCH(c, KC_QKV, s, launch_qkv_attn(g, x3, s));   // 93 % of its MFMA work is the in_proj GEMM
        } else if (pl.attn == AF_QKV_LONG) {
            // plain-bf16 phase, long sequence: in_proj + attention of one (sample, head) per workgroup, q / k / v stay in LDS
            QkvAttnArgs g{};
            g.Ahi = h_p.hi; g.a_rows = h_p.rows;
            g.Wfr = c->dp<__bf16>(f16 ? w.qkv.fr16 : w.qkv.fr); g.bias = c->dp<float>(w.qkv.b);
            g.out = att_p;
            g.Bm = ns; g.Kp = w.qkv.Kp; g.d = d; g.H = c->H; g.Tq = dm.Tq;
            g.qscale = 1.0f / sqrtf((float)dm.dh);
            g.f16 = f16 ? 1 : 0;
            RGN_LAUNCH(c, KC_QKV, s, launch_qkv_attn_long(g, s));
        } else if (pl.attn == AF_ROWGEMM_ATTN) {
            // plain-bf16 phase, long sequence: packed in_proj as a row-complete GEMM that scatters q (pre-scaled), k, v as
            // attention-ready planes (weights streamed to registers, output through an LDS image), then k_attn_x3
            RowGemmArgs g{};
            g.A = h_p.hi; g.a_rows = h_p.rows;
            g.W = c->dp<__bf16>(w.qkv.fr); g.bias = c->dp<float>(w.qkv.b);
            g.M = M; g.N = 3 * d; g.Kp = w.qkv.Kp; g.act = 2;
            g.Qhi = c->q_hi + slab0; g.Khi = c->k_hi + slab0; g.Vhi = c->vt_hi + slab0;
            g.H = c->H; g.dh = dm.dh; g.Tq = dm.Tq; g.Tqp = c->Tqp; g.qscale = 1.0f / sqrtf((float)dm.dh);
            RGN_LAUNCH(c, KC_ROWACT, s, launch_rowgemm(g, false, s));
            AttnX3Args a{};
            a.Qhi = g.Qhi; a.Qlo = c->q_lo + slab0; a.Khi = g.Khi; a.Klo = c->k_lo + slab0; a.Vthi = g.Vhi; a.Vtlo = c->vt_lo + slab0;
            a.out = att_p;
            a.Bm = ns; a.H = c->H; a.dh = dm.dh; a.d = d; a.Tq = dm.Tq; a.Tqp = c->Tqp; a.x3 = false;
            RGN_LAUNCH(c, KC_ATTN, s, launch_attn_x3(a, s));
        } else if (pl.attn == AF_GEMM_ATTN) {
            // in_proj GEMM scatters q (pre-scaled), k and v as attention-ready split planes; no fp32 qkv round trip
            GemmX3Args g{};
            g.Ahi = h_p.hi; g.Alo = h_p.lo; g.a_rows = h_p.rows;
            g.Whi = c->dp<__bf16>(w.qkv.hi); g.Wlo = c->dp<__bf16>(w.qkv.lo);
            g.bias = c->dp<float>(w.qkv.b);
            g.M = M; g.N = 3 * d; g.Kp = w.qkv.Kp;
            g.Qhi = c->q_hi + slab0; g.Qlo = x3 ? c->q_lo + slab0 : nullptr;
            g.Khi = c->k_hi + slab0; g.Klo = x3 ? c->k_lo + slab0 : nullptr;
            g.Vthi = c->vt_hi + slab0; g.Vtlo = x3 ? c->vt_lo + slab0 : nullptr;
            g.d = d; g.H = c->H; g.dh = dm.dh; g.Tq = dm.Tq; g.Tqp = c->Tqp;
            g.qscale = 1.0f / sqrtf((float)dm.dh);
            g.tq_magic = (unsigned)((1ull << 32) / (unsigned)dm.Tq) + 1u;
            RGN_LAUNCH(c, KC_GEMM, s, launch_gemm_x3(g, x3, 0, s));
            AttnX3Args a{};
            a.Qhi = g.Qhi; a.Qlo = c->q_lo + slab0; a.Khi = g.Khi; a.Klo = c->k_lo + slab0; a.Vthi = g.Vthi; a.Vtlo = c->vt_lo + slab0;
            a.out = att_p;
            a.Bm = ns; a.H = c->H; a.dh = dm.dh; a.d = d; a.Tq = dm.Tq; a.Tqp = c->Tqp; a.x3 = x3;
            RGN_LAUNCH(c, KC_ATTN, s, launch_attn_x3(a, s));
        } else {
            if ((rc = big(w.qkv, h, d, h_p, qkv, 3 * d, none, nullptr, 0, M))) return rc;
            RGN_LAUNCH(c, KC_ATTN, s, launch_attention(qkv, fast ? nullptr : att, att_p, dm, s));
        }
        const float* per_sample = sampling ? (ccond_rows ? ccond_rows + (size_t)s0 * Ld + (size_t)l * d : nullptr)
                                           : c->call + (size_t)s0 * Ld + (size_t)l * d;
        const float* step_vec = sampling ? c->call_time + (size_t)l * d : nullptr;
        if (pl.tail == TF_MLP_X3) {
            // split-bf16 phase, d = 512 / ff = 1024: the same layer tail on (hi, lo) plane pairs, three MFMAs per product (rgn_mlp_x3.hip):
            // one launch where k_gemm_x3 x 3 + k_layernorm x 2 were five; residual stream updated in place (both planes)
            MlpX3Args gx{};
            MlpArgs& g = gx.p;
            g.att = att_p.hi; g.h = h_p.hi; g.out = h_p.hi; g.rows = h_p.rows; g.M = M;
            gx.att_lo = att_p.lo; gx.h_lo = h_p.lo; gx.out_lo = h_p.lo;
            g.Wo = c->dp<__bf16>(w.out.fr); g.W1 = c->dp<__bf16>(w.ff1.fr); g.W2 = c->dp<__bf16>(w.ff2.fr);
            gx.Wo_lo = c->dp<__bf16>(w.out.fr_lo); gx.W1_lo = c->dp<__bf16>(w.ff1.fr_lo); gx.W2_lo = c->dp<__bf16>(w.ff2.fr_lo);
            g.bo = c->dp<float>(w.out.b); g.bf1 = c->dp<float>(w.ff1.b); g.bf2 = c->dp<float>(w.ff2.b);
            g.g1 = c->dp<float>(w.ln[0]); g.b1 = c->dp<float>(w.ln[1]); g.g2 = c->dp<float>(w.ln[2]); g.b2 = c->dp<float>(w.ln[3]);
            g.g3 = c->dp<float>(w.ln[4]); g.b3 = c->dp<float>(w.ln[5]);
            g.pervec = per_sample; g.ldper = Ld; g.stepvec = step_vec; g.ldstep = Ld; g.d_step = c->d_step; g.Tq = dm.Tq;
            RGN_LAUNCH(c, KC_MLP, s, launch_mlp_x3(gx, s));
            continue;
        }
        if (pl.tail == TF_MLP) {
            // plain-bf16 phase, d = 512 / ff = 1024: the whole layer tail (out_proj + norm1 + folded cross-attention + norm2 +
            // linear1 + GELU + linear2 + norm3) as ONE row-persistent kernel; residual stream updated in place (hi plane)
            MlpArgs g{};
            g.att = att_p.hi; g.h = h_p.hi; g.out = h_p.hi; g.rows = h_p.rows; g.M = M;
            g.Wo = c->dp<__bf16>(f16 ? w.out.fr16 : w.out.fr); g.W1 = c->dp<__bf16>(f16 ? w.ff1.fr16 : w.ff1.fr); g.W2 = c->dp<__bf16>(f16 ? w.ff2.fr16 : w.ff2.fr);
            g.f16 = f16 ? 1 : 0;
            g.bo = c->dp<float>(w.out.b); g.bf1 = c->dp<float>(w.ff1.b); g.bf2 = c->dp<float>(w.ff2.b);
            g.g1 = c->dp<float>(w.ln[0]); g.b1 = c->dp<float>(w.ln[1]); g.g2 = c->dp<float>(w.ln[2]); g.b2 = c->dp<float>(w.ln[3]);
            g.g3 = c->dp<float>(w.ln[4]); g.b3 = c->dp<float>(w.ln[5]);
            g.pervec = per_sample; g.ldper = Ld; g.stepvec = step_vec; g.ldstep = Ld; g.d_step = c->d_step; g.Tq = dm.Tq;
            RGN_LAUNCH(c, KC_MLP, s, launch_mlp(g, s));
            continue;
        }
        if (pl.tail == TF_ROWGEMM) {
            // plain-bf16 phase: out_proj + residual + norm1 + folded cross-attention + norm2 | linear1 + GELU |
            // linear2 + residual + norm3, three row-complete kernels; the residual stream is updated in place as planes
            RowGemmArgs g{};
            g.A = att_p.hi; g.a_rows = att_p.rows;
            g.W = c->dp<__bf16>(w.out.fr); g.bias = c->dp<float>(w.out.b);
            g.M = M; g.N = d; g.Kp = w.out.Kp;
            g.Rhi = h_p.hi; g.Rlo = h_p.lo; g.r_rows = h_p.rows; g.Ohi = h_p.hi; g.Olo = h_p.lo; g.o_rows = h_p.rows;
            g.ga = c->dp<float>(w.ln[0]); g.ba = c->dp<float>(w.ln[1]); g.gb = c->dp<float>(w.ln[2]); g.bb = c->dp<float>(w.ln[3]);
            g.pervec = per_sample; g.ldper = Ld; g.stepvec = step_vec; g.ldstep = Ld; g.d_step = c->d_step; g.Tq = dm.Tq;
            RGN_LAUNCH(c, KC_ROWLN, s, launch_rowgemm(g, true, s));
            RowGemmArgs f{};
            f.A = h_p.hi; f.a_rows = h_p.rows;
            f.W = c->dp<__bf16>(w.ff1.fr); f.bias = c->dp<float>(w.ff1.b);
            f.M = M; f.N = c->ff; f.Kp = w.ff1.Kp; f.act = 1;
            f.Chi = ffn_p.hi; f.Clo = ffn_p.lo; f.c_rows = ffn_p.rows;
            RGN_LAUNCH(c, KC_ROWACT, s, launch_rowgemm(f, false, s));
            g.A = ffn_p.hi; g.a_rows = ffn_p.rows;
            g.W = c->dp<__bf16>(w.ff2.fr); g.bias = c->dp<float>(w.ff2.b);
            g.Kp = w.ff2.Kp;
            g.ga = c->dp<float>(w.ln[4]); g.ba = c->dp<float>(w.ln[5]); g.gb = nullptr; g.bb = nullptr;
            g.pervec = nullptr; g.stepvec = nullptr;
            RGN_LAUNCH(c, KC_ROWLN, s, launch_rowgemm(g, true, s));
            continue;
        }
        if ((rc = big(w.out, att, d, att_p, tmp, d, none, h32 ? h : nullptr, 0, M))) return rc;
        RGN_LAUNCH(c, KC_LN, s,
                   launch_layernorm(tmp, h32 ? none : h_p, h32 ? h : nullptr, h_p, M, d, c->dp<float>(w.ln[0]), c->dp<float>(w.ln[1]), per_sample, Ld, step_vec, Ld,
                                    c->d_step, dm.Tq, c->dp<float>(w.ln[2]), c->dp<float>(w.ln[3]), s));
        if ((rc = big(w.ff1, h, d, h_p, fast ? nullptr : ffn, c->ff, ffn_p, nullptr, 1, M))) return rc;
        if ((rc = big(w.ff2, ffn, c->ff, ffn_p, tmp, d, none, h32 ? h : nullptr, 0, M))) return rc;
        RGN_LAUNCH(c, KC_LN, s,
                   launch_layernorm(tmp, h32 ? none : h_p, h32 ? h : nullptr, h_p, M, d, c->dp<float>(w.ln[4]), c->dp<float>(w.ln[5]), nullptr, 0, nullptr, 0,
                                    nullptr, dm.Tq, nullptr, nullptr, s));
    }
    if (c->skip_embed_out) return RGN_OK;   // (k_step applies the output projection)
    return big(c->lin_out, h, d, h_p, c->x0tok + (size_t)row0 * c->F, c->F, none, nullptr, 0, M);
}

// The input embedding of ALL rows into the residual-stream planes (hi): what every fused step leaves behind for the next
// one, needed once in front of the first fused step of a sampling call.
int embed_all(rgn_ctx* c, const Dims& dm, hipStream_t s) {
    const int M = dm.Bm * dm.Tq;
    GemmX3Args g{};
    g.Ahi = c->xin_hi; g.Alo = c->xin_lo; g.a_rows = M;
    g.Whi = c->dp<__bf16>(c->lin_x.hi); g.Wlo = c->dp<__bf16>(c->lin_x.lo);
    g.bias = nullptr;
    g.add = c->c0; g.ldadd = c->d;
    g.Chi = c->h_hi; g.Clo = nullptr; g.c_rows = M;
    g.M = M; g.N = c->d; g.Kp = c->lin_x.Kp;
    RGN_LAUNCH(c, KC_GEMM, s, launch_gemm_x3(g, false, (M >= c->big_tile_rows) ? 1 : 0, s));
    return RGN_OK;
}

// One denoiser evaluation on the bound condition, ending in k_update (sampler step or plain output).
// Everything t-dependent is read on the device (d_step / d_sp) so the sequence is graph-capturable.
int run_eval(rgn_ctx* c, int B, bool guided, bool uncond, bool sampling, hipStream_t s) {
    const Dims dm = make_dims(c, B, guided);
    const int prec = c->cfg.precision;
    const int d = c->d, Ld = c->L * c->d, M = dm.Bm * dm.Tq, Mb = B * dm.Tq;
    const EvalPlan pl = plan_eval(c, dm, guided, eval_x3(c), sampling);

    // timestep embedding (TimestepEmbedder cmdm.py:284-298) + condition embedding (cmdm.py:181-187).
    // Inside a sampling loop every sample shares t, so TE[s] and the folded cross-attention vectors were computed
    // once per schedule / condition (rgn_set_schedule, rgn_set_condition); rgn_denoise takes arbitrary per-sample
    // timesteps and evaluates them here.
    const bool has_cond = c->cfg.cond_mode != RGN_COND_NONE;
    const float* cond_rows = !has_cond ? nullptr : ((uncond && !guided) ? c->condemb + (size_t)B * d : c->condemb);
    const float* ccond_rows = !has_cond ? nullptr : ((uncond && !guided) ? c->call_cond + (size_t)B * Ld : c->call_cond);
    if (!sampling) {
        RGN_LAUNCH(c, KC_EMBED, s, launch_gather_pe(c->dp<float>(c->off_pe), c->d_tab, c->d_step, c->d_sp, c->pe_rows, dm.Bm, B, d, c->pe_len, s));
        GemmArgs g = gemm_args(c, c->lin_t0, c->pe_rows, d, c->emb1, d, dm.Bm);
        g.act = 2;
        RGN_LAUNCH(c, KC_GEMM, s, launch_gemm(g, small_prec(c), s));
        g = gemm_args(c, c->lin_t2, c->emb1, d, c->emb, d, dm.Bm);
        g.add = cond_rows;
        g.ldadd = d;
        RGN_LAUNCH(c, KC_GEMM, s, launch_gemm(g, small_prec(c), s));
        // cross-attention onto the 1-token memory, all layers at once: call[b, l*d:(l+1)*d]
        g = gemm_args(c, c->lin_g, c->emb, d, c->call, Ld, dm.Bm);
        RGN_LAUNCH(c, KC_GEMM, s, launch_gemm(g, small_prec(c), s));
    }
    // ---- layers. In the bf16 modes the samples of the evaluation are split into contiguous groups that run as
    //      independent kernel chains on separate streams (fork/join with events, also inside graph capture): the
    //      MFMA-bound GEMM main loops of one chain overlap the HBM-bound phases (GEMM epilogues, LayerNorm, attention)
    //      of the others. Samples are independent, so no kernel ever looks across a split.
    const bool fast = prec != RGN_PREC_F32;
    int rc;
    int nch = (fast && !c->prof) ? c->nchains : 1;       // per-kernel event timing wants un-overlapped kernels
    // Two chains instead of four when the whole evaluation is 129 .. 256 row tiles of 64 (B=256 at 60 frames: 240): each of the
    // two chains' launches then still fills half the chip in one round, with half the launches and joins (measured 314.4 vs
    // 308.9 motions/s at cfg2 in round 2, when larger evaluations - cfg3 480, cfg4 300 tiles - lost 5-6 % with two chains; smaller ones keep four)
    // (plain-bf16 phase only: the split-bf16 kernels - 128-row tiles, separate LayerNorms - measure 107 vs 125 motions/s with two)
    // Round 3 (one-sample attention workgroups, write-through stores): 385 .. 512 tiles (cfg3: 480 = two chains of one full round each)
    // now also prefer two - 1899 vs 1869 motions/s, three 1842, six 1581; cfg4's 300 tiles keep four (969 / 928 / 913 with 2 / 3 / 4),
    // cfg5's 1200 measure the same with two, three and four.
    const int tiles64 = (M + 63) / 64;
    if (nch == 4 && !c->nchains_user && !eval_x3(c) && ((tiles64 > 128 && tiles64 <= 256) || (tiles64 > 384 && tiles64 <= 512))) nch = 2;
    // up to 48 tiles: ONE chain, in both phases (B = 32 at 60 frames, 250-step calls: plain-bf16 phase 108.3 vs 114.1 ms with four chains,
    // split-bf16 phase 246 vs 277; one chain in the bulk phase and four in the tail measured 120 - 137 ms against 111 with one throughout)
    // (B = 40 / 48: 114.3 / 113.8 vs 120.0 / 118.8 with four; B = 64, 60 tiles: the same with one, two and four)
    if (nch == 4 && !c->nchains_user && tiles64 <= 48) nch = 1;
    if (pl.sb) nch = 1;                                   // small-batch engine: one chain of column-split kernels
    if (nch > dm.Bm) nch = dm.Bm;
    if (nch > 1) RGN_HIP(c, hipEventRecord(c->ev_fork, s));
    const int per = dm.Bm / nch, extra = dm.Bm % nch;
    int s0 = per + (extra > 0 ? 1 : 0);                   // chain 0 (main stream) takes [0, s0) and is enqueued last
    const int first_n = s0;
    // Without guidance a chain's samples are all the update kernel of that chain needs, so it runs at the end of the
    // chain (overlapping the other chains' layers); with guidance the cond / uncond halves of a sample sit in different
    // chains and the update waits for the join.
    const Planes xin_p{fast ? c->xin_hi : nullptr, (fast && has_lo(c)) ? c->xin_lo : nullptr, M};
    c->skip_embed_out = false;
    const bool fused = pl.step_fused;                       // k_step instead of out GEMM + k_update + next in GEMM
    const bool own_update = !guided && (nch > 1 || fused);
    int total_tiles = 0;
    if (fused) {
        if (guided) total_tiles = (Mb + 63) / 64;            // one launch over the conditional rows, after the join
        else for (int k2 = 0; k2 < nch; ++k2) total_tiles += ((per + (k2 < extra ? 1 : 0)) * dm.Tq + 63) / 64;
        c->skip_embed_out = true;
    }
    auto step_or_update = [&](int s_first, int n, hipStream_t st) -> int {
        if (fused) {
            StepArgs g{};
            const size_t row0 = (size_t)s_first * dm.Tq;
            g.h = c->h_hi + row0 * 32; g.hout = c->h_hi + row0 * 32; g.rows = M; g.M = n * dm.Tq;
            const bool f16 = c->phase_f16 && !eval_x3(c);
            g.Wout = c->dp<__bf16>(f16 ? c->lin_out.fr16 : c->lin_out.fr); g.bout = c->dp<float>(c->lin_out.b); g.F = c->F; g.nb_out = (c->F + 31) / 32;
            g.Wx = c->dp<__bf16>(f16 ? c->lin_x.fr16 : c->lin_x.fr); g.nkx = c->lin_x.Kp / 32;
            g.c0 = (f16 ? reinterpret_cast<const __bf16*>(c->c0h16) : c->c0h) + row0 * c->d;
            g.f16 = f16 ? 1 : 0;
            g.tab = c->d_tab; g.d_step = c->d_step; g.sp = c->d_sp;
            g.T = dm.T; g.B = dm.B; g.s0 = s_first; g.total_tiles = total_tiles; g.no_quads = c->step_no_quads;
            if (guided) { g.scale = c->scale; g.half = Mb; }  // x0 = x0_u + scale (x0_c - x0_u); rows [Mb, 2 Mb) are the unconditional half
            RGN_LAUNCH(c, KC_STEP, st, launch_step(g, st));
        } else {
            RGN_LAUNCH(c, KC_UPDATE, st, launch_update(c->x0tok, c->scale, c->d_tab, c->d_step, c->d_sp, nullptr, xin_p, dm, s_first, n, st));
        }
        return RGN_OK;
    };
    for (int k = 1; k < nch; ++k) {
        const int n = per + (k < extra ? 1 : 0);
        RGN_HIP(c, hipStreamWaitEvent(c->side[k - 1], c->ev_fork, 0));
        if ((rc = run_layers(c, dm, guided, sampling, cond_rows, ccond_rows, s0, n, c->side[k - 1]))) return rc;
        if (own_update && (rc = step_or_update(s0, n, c->side[k - 1]))) return rc;
        RGN_HIP(c, hipEventRecord(c->ev_join[k - 1], c->side[k - 1]));
        s0 += n;
    }
    if ((rc = run_layers(c, dm, guided, sampling, cond_rows, ccond_rows, 0, first_n, s))) return rc;
    if (own_update && (rc = step_or_update(0, first_n, s))) return rc;
    c->skip_embed_out = false;
    for (int k = 1; k < nch; ++k) RGN_HIP(c, hipStreamWaitEvent(s, c->ev_join[k - 1], 0));
    if (fused && guided) {
        if ((rc = step_or_update(0, dm.B, s))) return rc;
    } else if (!own_update)
        RGN_LAUNCH(c, KC_UPDATE, s,
                   launch_update(c->x0tok, c->scale, c->d_tab, c->d_step, c->d_sp, fast ? nullptr : c->xin, xin_p, dm, 0, dm.B, s));
    return RGN_OK;
}

// fp32 emulation of the scalar arithmetic of p_sample / ddim_sample (gaussian_diffusion.py:544-559,
// 771-793): every table entry is cast fp64->fp32 first (_extract_into_tensor), then combined in fp32.
int build_step_table(rgn_ctx* c, float eta) {
    if (c->tab_valid && c->tab_eta == eta) return RGN_OK;
    std::vector<StepCoef> tab(c->S);
    for (int i = 0; i < c->S; ++i) {
        StepCoef k{};
        const float nz = (i != 0) ? 1.f : 0.f;
        k.c1 = (float)c->coef1[i];
        k.c2 = (float)c->coef2[i];
        volatile float half_lv = 0.5f * (float)c->logvar[i];
        k.sig_ddpm = nz * expf(half_lv);
        k.sr = (float)c->srecip[i];
        k.srm1 = (float)c->srecipm1[i];
        const float ab = (float)c->ac[i], abp = (float)c->acp[i];
        volatile float r1 = (1.f - abp) / (1.f - ab);
        volatile float r2 = 1.f - ab / abp;
        volatile float s1 = sqrtf(r1), s2 = sqrtf(r2);
        volatile float sig0 = eta * s1;
        volatile float sigma = sig0 * s2;
        k.ca = sqrtf(abp);
        volatile float sg2 = sigma * sigma;
        volatile float inner = 1.f - abp;
        inner = inner - sg2;
        k.cb = sqrtf(inner);
        k.sig_ddim = nz * sigma;
        k.t_model = (int32_t)c->tmap[i];
        tab[i] = k;
    }
    // earlier sampling calls may still be reading the table on the engine's (non-blocking) stream
    RGN_HIP(c, hipStreamSynchronize(c->stream));
    RGN_HIP(c, hipMemcpy(c->d_tab, tab.data(), tab.size() * sizeof(StepCoef), hipMemcpyHostToDevice));
    c->tab_eta = eta;
    c->tab_valid = true;
    return RGN_OK;
}

int sample_range(rgn_ctx* c, int32_t sampler, int32_t guided, float eta, float* x, const float* noise, uint64_t seed, uint64_t sample_offset,
                 int32_t first_index, int32_t count, float* x0_out, int32_t use_graph, int32_t clip_denoised, void* stream) {
    if (!c->have_sched) return c->fail(RGN_ERR_STATE, "rgn_sample_range: no schedule (rgn_set_schedule)");
    if (!c->have_cond) return c->fail(RGN_ERR_STATE, "rgn_sample_range: no condition bound (rgn_set_condition)");
    if (!x) return c->fail(RGN_ERR_INVALID_ARG, "rgn_sample_range: null x");
    if (sampler != RGN_SAMPLER_DDPM && sampler != RGN_SAMPLER_DDIM) return c->fail(RGN_ERR_INVALID_ARG, "rgn_sample_range: sampler");
    if (count <= 0 || first_index >= c->S || first_index - count + 1 < 0)
        return c->fail(RGN_ERR_INVALID_ARG, "rgn_sample_range: step range outside [0, S)");
    if (guided && c->cfg.cond_mode == RGN_COND_NONE)
        return c->fail(RGN_ERR_INVALID_ARG, "rgn_sample_range: guidance needs cond_mode text/action (cfg_sampler.py:26)");
    if (guided && !c->cond_has_scale) return c->fail(RGN_ERR_STATE, "rgn_sample_range: guided sampling needs y['scale']");
    hipStream_t us = reinterpret_cast<hipStream_t>(stream), s = c->stream;
    RGN_HIP(c, hipSetDevice(c->cfg.device));
    int rc = build_step_table(c, eta);
    if (rc) return rc;
    if ((rc = stream_enter(c, us))) return rc;
    const Dims dm = make_dims(c, c->B, guided != 0);
    SampleParams sp{};
    sp.x = x;
    sp.noise = noise;
    sp.x0_out = x0_out;
    sp.t_ext = nullptr;
    sp.seed = seed;
    sp.sample_offset = sample_offset;
    sp.first_index = first_index;
    sp.sampler = sampler;
    sp.mode = 0;
    sp.guided = guided != 0;
    sp.clip = clip_denoised != 0;
    sp.const_noise = c->const_noise;
    RGN_HIP(c, hipMemcpyAsync(c->d_sp, &sp, sizeof(sp), hipMemcpyHostToDevice, s));
    RGN_HIP(c, hipMemcpyAsync(c->d_step, &first_index, sizeof(int), hipMemcpyHostToDevice, s));
    RGN_HIP(c, hipMemsetAsync(c->d_step + 4, 0, (size_t)(1 + c->cfg.max_batch) * sizeof(int), s));   // k_update's ticket counters (clean even after an aborted call)
    if ((rc = pack_state(c, x, dm, guided != 0, s))) return rc;

    // Precision schedule: loop indices >= tail run the plain-bf16 phase, the last `tail` indices the split-bf16 one.
    // One captured step graph per phase; everything t-dependent is read on the device, so each serves all its steps.
    const bool sched = c->cfg.precision == RGN_PREC_BF16_X3TAIL;
    const PrecPlan pp = prec_plan(c, dm, guided != 0);
    const int tail = pp.tail, n16 = pp.n16;
    // A graph holds `steps` consecutive loop iterations (evaluation + sampler update + counter decrement each): the loop
    // index lives on the device, so one instantiated graph serves any starting index. Long ranges replay the multi-step
    // graph (graph_steps iterations per host launch; a 4-branch launch costs the host ~1 ms, as much as the GPU needs for
    // a step at B = 256), the remainder single-step graphs.
    auto graph_for = [&](bool x3, bool f16g, int steps, hipGraphExec_t* out) -> int {
        const uint64_t key = (uint64_t)c->B | ((uint64_t)(guided != 0) << 20) | ((uint64_t)sampler << 21) | ((uint64_t)x3 << 23) |
                             ((uint64_t)steps << 24) | ((uint64_t)f16g << 40);
        auto it = c->graphs.find(key);
        if (it != c->graphs.end()) {
            *out = it->second;
            return RGN_OK;
        }
        hipGraph_t graph = nullptr;
        hipGraphExec_t ge = nullptr;
        c->phase_x3 = x3;
        c->phase_f16 = f16g;
        RGN_HIP(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int r = RGN_OK;
        for (int k = 0; k < steps && r == RGN_OK; ++k) {
            r = run_eval(c, c->B, guided != 0, false, true, s);   // (its k_update also moves the device-side loop index on)
        }
        hipError_t e = hipStreamEndCapture(s, &graph);
        if (r) {
            if (graph) (void)hipGraphDestroy(graph);
            return r;
        }
        RGN_HIP(c, e);
        RGN_HIP(c, hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        c->graphs[key] = ge;
        *out = ge;
        return RGN_OK;
    };
    // Graph replay is what the throughput engine needs (a step is 2-4 concurrent kernel chains the host could not feed: eager
    // launches are 2x slower at B = 16). The small-batch engine is one chain of ~43 short kernels per step, and there every graph
    // node costs ~0.4 us more than the same kernel launched from this loop (B = 1: 278 vs 261 ms per 1000 steps, B = 4: 351 vs 338,
    // B = 12: 492 vs 487; the host needs ~150 ms per 1000 steps to issue them): it launches eagerly unless REGENNET_SB_GRAPH is set.
    const bool sb_graph = opt_flag(c, "SB_GRAPH");
    const bool graphs = use_graph && !c->prof && (sb_graph || !use_sb(c, dm.Bm * dm.Tq));
    const int multi = c->graph_steps;
    int k = 0;
    // Fused step boundaries (k_step) hand the next evaluation's input embedding over in the residual-stream planes and no
    // longer write the token-major x planes: the first fused step of the call needs the embedding made once up front, and
    // the first un-fused step behind fused ones (the split-bf16 tail) needs the planes re-made from the sampler state.
    bool prev_fused = false, planes_f16 = false;
    while (k < count) {
        const int i = first_index - k;                       // loop index of the next step
        const bool x3 = !sched || i < tail;
        const bool f16 = !x3 && i < tail + n16;              // (n16 > 0 only where the plain phase is k_layers<true>)
        const int phase_end = x3 ? 0 : (f16 ? tail : tail + n16);            // first loop index behind this phase
        const int phase_left = (i - phase_end + 1) < (count - k) ? (i - phase_end + 1) : (count - k);              // steps left in this phase
        const EvalPlan pl = plan_eval(c, dm, guided != 0, x3, true);
        const bool fused_now = pl.step_fused;
        if (fused_now && !prev_fused) {
            if ((rc = embed_all(c, dm, s))) return rc;
            planes_f16 = false;
        }
        if (f16 && !planes_f16) {   // the fp16-operand forms read (and rewrite) the residual-stream planes as fp16: what the embedding or the bf16 steps left there is re-encoded
            RGN_LAUNCH(c, KC_EMBED, s, launch_bf16_to_f16(c->h_hi, (size_t)dm.Bm * dm.Tq * c->d, s));
            planes_f16 = true;
        }
        if (!fused_now && prev_fused && (rc = pack_state(c, x, dm, guided != 0, s))) return rc;
        prev_fused = fused_now;
        if (pl.steps) {
            // plain-bf16 phase, <= 64 tokens: ALL remaining steps of the phase in one launch - a workgroup carries its sample (guided: its
            // motion's two evaluations) through decoder stack and step boundary step after step; nothing but x, the condition rows and the
            // weights is read
            {
                const int M = dm.Bm * dm.Tq;
                const bool has_cond = c->cfg.cond_mode != RGN_COND_NONE;
                LayersArgs g{};
                g.h = c->h_hi; g.out = c->h_hi; g.rows = M; g.Bm = dm.B;   // one workgroup per MOTION (guided: its two evaluations back to back)
                fill_layers_args(c, g, dm, true, has_cond ? c->call_cond : nullptr, 0, f16);
                g.steps = phase_left;
                g.f16 = f16 ? 1 : 0;
                if (guided) {
                    g.scale = c->scale; g.half = dm.B * dm.Tq;
                    g.park = reinterpret_cast<float*>(c->ffn_hi);           // (the hidden-tensor planes are idle on this path: 2B * T * ff * 2 bytes >= B * 96 KiB)
                }
                g.Wout = c->dp<__bf16>(f16 ? c->lin_out.fr16 : c->lin_out.fr); g.bout = c->dp<float>(c->lin_out.b); g.F = c->F; g.nb_out = (c->F + 31) / 32;
                g.Wx = c->dp<__bf16>(f16 ? c->lin_x.fr16 : c->lin_x.fr);
                g.c0 = f16 ? reinterpret_cast<const __bf16*>(c->c0h16) : c->c0h;
                g.tab = c->d_tab; g.d_stepw = c->d_step; g.sp = c->d_sp;
                g.B = dm.B; g.s0 = 0; g.no_quads = c->step_no_quads;
                RGN_LAUNCH(c, KC_STEPS, s, launch_layers(g, s));
                k += phase_left;
                continue;
            }
        }
        if (graphs) {
            const int steps = (multi > 1 && phase_left >= multi) ? multi : 1;
            hipGraphExec_t ge = nullptr;
            if ((rc = graph_for(x3, f16, steps, &ge))) return rc;
            RGN_HIP(c, hipGraphLaunch(ge, s));
            k += steps;
        } else {
            c->phase_x3 = x3;
            c->phase_f16 = f16;
            rc = run_eval(c, c->B, guided != 0, false, true, s);
            if (rc) return rc;
            k += 1;
        }
    }
    c->phase_x3 = true;
    c->phase_f16 = false;
    return stream_exit(c, us);
}

int plan_query(rgn_ctx* c, int32_t B, int32_t guided, int32_t split_phase, int32_t idx, const char** name, const char** kernel,
               double* launches_per_eval, double* algo_flops_per_eval, double* l2_bytes_per_eval) {
    if (idx < 0 || idx >= KC_COUNT || !name || !kernel || !launches_per_eval || !algo_flops_per_eval || !l2_bytes_per_eval)
        return c->fail(RGN_ERR_INVALID_ARG, "rgn_plan_query: bad argument");
    if (!c->finalized) return c->fail(RGN_ERR_STATE, "rgn_plan_query: weights not finalized");
    if (B <= 0 || B > c->cfg.max_batch) return c->fail(RGN_ERR_INVALID_ARG, "rgn_plan_query: B outside (0, max_batch]");
    const Dims dm = make_dims(c, B, guided != 0);
    const bool x3 = eval_x3_phase(c, split_phase != 0);
    const EvalPlan pl = plan_eval(c, dm, guided != 0, x3, true);
    // SURVEY.md 8(d) accounting: MACs of ONE evaluation of the bound batch (2 B rows under guidance), full T x T attention scores; the
    // timestep MLP and the folded 1-token cross-attention are per-schedule / per-condition work, not per step
    const double T = dm.T, d = c->d, ff = c->ff, L = c->L, F = c->F, M = (double)dm.Bm * T;
    const double qkv = M * 3 * d * d * L, attn = M * 2 * T * d * L, tail = M * (d * d + 2 * d * ff) * L;
    const double embed = (c->cfg.precision == RGN_PREC_F32 ? (double)dm.B * T * F * d : M * F * d) + M * d * F;   // input embedding + output projection
    double mac[KC_COUNT] = {0}, n[KC_COUNT] = {0}, l2[KC_COUNT] = {0};
    const char* kn[KC_COUNT] = {nullptr};
    for (int i = 0; i < KC_COUNT; ++i) kn[i] = "";
    const double Fp = (double)align_up((size_t)c->F, 32), wl = L * (4 * d * d + 2 * d * ff);
    if (pl.sb) {
        const bool fa = pl.attn == AF_QKV;
        mac[KC_SB] = embed + tail + (fa ? 0.0 : qkv); n[KC_SB] = 2 + L * (fa ? 3 : 4); kn[KC_SB] = "k_sb_gemm";
        if (fa) { mac[KC_QKV] = qkv + attn; n[KC_QKV] = L; kn[KC_QKV] = "k_sb_qkv_attn"; }
        else { mac[KC_ATTN] = attn; n[KC_ATTN] = L; kn[KC_ATTN] = "k_attn_x3"; }
        n[KC_UPDATE] = 1; kn[KC_UPDATE] = "k_update";
    } else {
        const bool f32 = c->cfg.precision == RGN_PREC_F32;
        const char* gemm = f32 ? "k_gemm_f32" : "k_gemm_x3";
        kn[KC_GEMM] = gemm;
        if (pl.steps) {
            mac[KC_STEPS] = qkv + attn + tail + embed; {
                const PrecPlan pp = c->have_sched ? prec_plan(c, dm, guided != 0) : PrecPlan{};
                const bool all16 = c->have_sched && pp.n16 > 0 && pp.n16 >= c->S - pp.tail;   // every plain step of the bound schedule runs on fp16 operands
                kn[KC_STEPS] = all16 ? (guided ? "k_layers<true, true, f16>" : "k_layers<true, false, f16>") : (guided ? "k_layers<true, true>" : "k_layers<true>");
            }
            n[KC_STEPS] = 0;   // ONE launch per run of steps (rgn_sample_range), not per evaluation
            const double passes = guided ? 2 : 1;
            l2[KC_STEPS] = (double)dm.B * (passes * (wl + Fp * d) + Fp * d) * 2.0;
        } else {
            if (pl.layers) {
                mac[KC_LAYERS] = qkv + attn + tail; n[KC_LAYERS] = 1; kn[KC_LAYERS] = "k_layers<false>";
                l2[KC_LAYERS] = (double)dm.Bm * wl * 2.0;
            } else {
                switch (pl.attn) {
                case AF_QKV: mac[KC_QKV] = qkv + attn; n[KC_QKV] = L; kn[KC_QKV] = !c->qkv_rs ? "k_qkv_attn" : x3 ? (c->qkv_x3_dma || c->d != 512 ? "k_qkv_attn" : "k_qkv_attn_rs_x3") : "k_qkv_attn_rs"; break;
                case AF_QKV_LONG: mac[KC_QKV] = qkv + attn; n[KC_QKV] = L; kn[KC_QKV] = "k_qkv_attn_long"; break;
                case AF_ROWGEMM_ATTN: mac[KC_ROWACT] += qkv; n[KC_ROWACT] += L; mac[KC_ATTN] = attn; n[KC_ATTN] = L; kn[KC_ATTN] = "k_attn_x3"; break;
                case AF_GEMM_ATTN: mac[KC_GEMM] += qkv; n[KC_GEMM] += L; mac[KC_ATTN] = attn; n[KC_ATTN] = L; kn[KC_ATTN] = "k_attn_x3"; break;
                default: mac[KC_GEMM] += qkv; n[KC_GEMM] += L; mac[KC_ATTN] = attn; n[KC_ATTN] = L; kn[KC_ATTN] = f32 ? "k_attn_mfma" : "k_attention"; break;
                }
                switch (pl.tail) {
                case TF_MLP_X3: mac[KC_MLP] = tail; n[KC_MLP] = L; kn[KC_MLP] = "k_mlp_x3"; break;
                case TF_MLP: mac[KC_MLP] = tail; n[KC_MLP] = L; kn[KC_MLP] = "k_mlp2"; break;
                case TF_ROWGEMM:
                    mac[KC_ROWLN] = M * (d * d + d * ff) * L; n[KC_ROWLN] = 2 * L; kn[KC_ROWLN] = "k_rowgemm<LN>";
                    mac[KC_ROWACT] += M * d * ff * L; n[KC_ROWACT] += L;
                    break;
                default: mac[KC_GEMM] += tail; n[KC_GEMM] += 3 * L; n[KC_LN] = 2 * L; kn[KC_LN] = "k_layernorm"; break;
                }
                kn[KC_ROWACT] = "k_rowgemm<ACT>";
            }
            if (pl.step_fused) { mac[KC_STEP] = embed; n[KC_STEP] = 1; kn[KC_STEP] = guided ? "k_step<guided>" : "k_step"; }
            else { mac[KC_GEMM] += embed; n[KC_GEMM] += 2; n[KC_UPDATE] = 1; kn[KC_UPDATE] = "k_update"; }
        }
    }
    *name = kclass_names[idx];
    *kernel = kn[idx];
    *launches_per_eval = n[idx];
    *algo_flops_per_eval = 2.0 * mac[idx];
    *l2_bytes_per_eval = l2[idx];
    return RGN_OK;
}

}  // namespace rgnh
