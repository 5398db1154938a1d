// Checkpoint ingestion (host): the reference's state_dict (key names and shapes of model/cmdm.py:53-105, checked in rgn_load_weight) is folded
// (fuse_process into the pose embeddings, the 1-token cross-attention into G = Wo Wv: fp64) and packed into ONE 256-byte-aligned device blob -
// fp32 [N, Kp], bf16 hi / lo planes (row-major or K32-blocked), MFMA-fragment-ordered bf16 hi / lo and fp16 planes - and the activation workspace
// is allocated. utils/model_util.py:5-8 (load_model_wo_clip) is the reference interface this serves.
#include "rgn_host.h"

namespace rgnh {

void build_expected(rgn_ctx* c) {
    const int64_t d = c->d, F = c->F, ff = c->ff;
    auto& e = c->expected;
    e["input_process.poseEmbedding.weight"] = {d, F};
    e["input_process.poseEmbedding.bias"] = {d};
    e["cmo_process.poseEmbedding.weight"] = {d, F};
    e["cmo_process.poseEmbedding.bias"] = {d};
    if (c->cfg.cm_mode == RGN_CM_CONCAT) {
        e["fuse_process.weight"] = {d, 2 * d};
        e["fuse_process.bias"] = {d};
    }
    e["sequence_pos_encoder.pe"] = {-1, 1, d};
    e["embed_timestep.sequence_pos_encoder.pe"] = {-1, 1, d};
    e["embed_timestep.time_embed.0.weight"] = {d, d};
    e["embed_timestep.time_embed.0.bias"] = {d};
    e["embed_timestep.time_embed.2.weight"] = {d, d};
    e["embed_timestep.time_embed.2.bias"] = {d};
    for (int l = 0; l < c->L; ++l) {
        const std::string p = "seqTransDecoder.layers." + std::to_string(l) + ".";
        for (const char* a : {"self_attn.", "multihead_attn."}) {
            e[p + a + "in_proj_weight"] = {3 * d, d};
            e[p + a + "in_proj_bias"] = {3 * d};
            e[p + a + "out_proj.weight"] = {d, d};
            e[p + a + "out_proj.bias"] = {d};
        }
        e[p + "linear1.weight"] = {ff, d};
        e[p + "linear1.bias"] = {ff};
        e[p + "linear2.weight"] = {d, ff};
        e[p + "linear2.bias"] = {d};
        for (const char* n : {"norm1", "norm2", "norm3"}) {
            e[p + n + ".weight"] = {d};
            e[p + n + ".bias"] = {d};
        }
    }
    if (c->cfg.cond_mode == RGN_COND_TEXT) {
        e["embed_text.weight"] = {d, c->cfg.clip_dim};
        e["embed_text.bias"] = {d};
    }
    if (c->cfg.cond_mode == RGN_COND_ACTION) e["embed_action.action_embedding"] = {c->cfg.num_actions, d};
    e["output_process.poseFinal.weight"] = {F, d};
    e["output_process.poseFinal.bias"] = {F};
}

// ---- blob building --------------------------------------------------------------------------------
static size_t blob_put(rgn_ctx* c, const void* src, size_t bytes) {
    const size_t off = align_up(c->hblob.size(), 256);
    c->hblob.resize(off + bytes);
    if (src) memcpy(c->hblob.data() + off, src, bytes);
    return off;
}

// Pack W[N,K] (row-major fp32) into fp32 [N,Kp] plus bf16 hi/lo planes; bias optional.
static Lin pack_linear(rgn_ctx* c, const float* W, const float* bias, int N, int K, bool blocked = false, bool frag = false, bool frag_lo = false, bool frag16 = false) {
    Lin L;
    L.blocked = blocked;
    L.N = N;
    L.K = K;
    L.Kp = (int)align_up((size_t)K, 32);
    std::vector<float> w((size_t)N * L.Kp, 0.f);
    std::vector<uint16_t> hi((size_t)N * L.Kp, 0), lo((size_t)N * L.Kp, 0);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const float v = W[(size_t)n * K + k];
            const size_t o = (size_t)n * L.Kp + k;
            const size_t ob = blocked ? ((size_t)(k / 32) * N + n) * 32 + (k % 32) : o;
            w[o] = v;
            hi[ob] = f2bf(v);
            lo[ob] = f2bf(v - bf2f(hi[ob]));
        }
    L.w = blob_put(c, w.data(), w.size() * 4);
    L.hi = blob_put(c, hi.data(), hi.size() * 2);
    L.lo = blob_put(c, lo.data(), lo.size() * 2);
    if (frag) {
        // fragment order [Kp/32][Np/32][ks 2][lane 64][8]: lane = 32 * ((k % 16) / 8) + n % 32 holds its 8 consecutive k
        // (rows zero-padded to Np = a multiple of 32: only the output projection, N = F, needs it)
        const size_t Np = align_up((size_t)N, 32);
        std::vector<uint16_t> fr(Np * L.Kp, 0);
        const size_t nb_all = Np / 32;
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k) {
                const size_t kt = k / 32, ks = (k % 32) / 16, lane = 32 * ((k % 16) / 8) + n % 32;
                fr[(((kt * nb_all + n / 32) * 2 + ks) * 64 + lane) * 8 + k % 8] = f2bf(W[(size_t)n * K + k]);
            }
        L.fr = blob_put(c, fr.data(), fr.size() * 2);
        if (frag_lo) {   // the lo plane of the split, same order: with fr the operand pair of k_mlp_x3
            std::vector<uint16_t> fl(Np * L.Kp, 0);
            for (int n = 0; n < N; ++n)
                for (int k = 0; k < K; ++k) {
                    const size_t kt = k / 32, ks = (k % 32) / 16, lane = 32 * ((k % 16) / 8) + n % 32;
                    const float v = W[(size_t)n * K + k];
                    fl[(((kt * nb_all + n / 32) * 2 + ks) * 64 + lane) * 8 + k % 8] = f2bf(v - bf2f(f2bf(v)));
                }
            L.fr_lo = blob_put(c, fl.data(), fl.size() * 2);
        }
        if (frag16) {    // the same plane as IEEE fp16 (k_layers' fp16-operand form)
            std::vector<uint16_t> fh(Np * L.Kp, 0);
            for (int n = 0; n < N; ++n)
                for (int k = 0; k < K; ++k) {
                    const size_t kt = k / 32, ks = (k % 32) / 16, lane = 32 * ((k % 16) / 8) + n % 32;
                    fh[(((kt * nb_all + n / 32) * 2 + ks) * 64 + lane) * 8 + k % 8] = f2h(W[(size_t)n * K + k]);
                }
            L.fr16 = blob_put(c, fh.data(), fh.size() * 2);
        }
    }
    if (bias) {
        L.b = blob_put(c, bias, (size_t)N * 4);
        L.has_bias = true;
    }
    return L;
}

// C[n,k] = sum_j A[n,j] * B[j,k]  in fp64 (weight folding at load time)
static void matmul64(const float* A, const float* Bm, int N, int J, int K, std::vector<double>& C) {
    C.assign((size_t)N * K, 0.0);
    for (int n = 0; n < N; ++n) {
        double* cr = &C[(size_t)n * K];
        for (int j = 0; j < J; ++j) {
            const double a = A[(size_t)n * J + j];
            const float* br = Bm + (size_t)j * K;
            for (int k = 0; k < K; ++k) cr[k] += a * (double)br[k];
        }
    }
}

template <typename T>
static int ws_alloc(rgn_ctx* c, T** p, size_t count) {
    void* q = nullptr;
    RGN_HIP(c, hipMalloc(&q, count * sizeof(T) + 256));
    c->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return RGN_OK;
}

// A kernel-selection switch of this handle: rgn_set_option(h, "KEY", v) if given, else the environment variable REGENNET_KEY, else absent.
// (The environment stays as the process-wide default - tools, A/B runs; a library user or a test addresses ONE handle.)
bool opt_get(const rgn_ctx* c, const char* key, int* value) {
    auto it = c->opts.find(key);
    if (it != c->opts.end()) {
        *value = it->second;
        return true;
    }
    const std::string env = std::string("REGENNET_") + key;
    if (const char* e = getenv(env.c_str())) {
        *value = atoi(e);
        return true;
    }
    return false;
}
// "flag" switches (REGENNET_NO_MLP ...): on when the variable exists at all / when the option was set to a non-zero value
bool opt_flag(const rgn_ctx* c, const char* key) {
    auto it = c->opts.find(key);
    if (it != c->opts.end()) return it->second != 0;
    const std::string env = std::string("REGENNET_") + key;
    return getenv(env.c_str()) != nullptr;
}

int finalize_weights(rgn_ctx* c) {
    std::string missing;
    for (auto& kv : c->expected)
        if (!c->sd.count(kv.first)) missing += (missing.empty() ? "" : ", ") + kv.first;
    if (!missing.empty()) return c->fail(RGN_ERR_MISSING_KEY, "missing keys in state_dict: " + missing);
    RGN_HIP(c, hipSetDevice(c->cfg.device));
    auto W = [&](const std::string& k) -> const float* { return c->sd[k].v.data(); };
    const int d = c->d, F = c->F, ff = c->ff;
    // fp16 operands for the plain phase of the precision schedule (k_layers<.., F16>): same MFMA rate and bytes as bf16, 2^-12 instead of
    // 2^-9 operand rounding - but a 5-bit exponent: a weight at or beyond fp16's range would become inf, so such a checkpoint is refused
    // here, with the key named (activations are LayerNorm outputs, probabilities and GELU values: O(1) by construction)
    { int v = 1; (void)opt_get(c, "BULK_F16", &v); c->bulk_f16 = c->cfg.precision == RGN_PREC_BF16_X3TAIL && v != 0; }
    { int v; if (opt_get(c, "F16_STEPS", &v)) c->f16_steps = c->f16_steps_default = v < 0 ? -1 : v; }
    auto f16_range = [&](const std::string& key, const float* w, size_t n) -> bool {
        for (size_t i = 0; i < n; ++i)
            if (!(std::fabs(w[i]) < 6.0e4f)) {
                c->hblob.clear();   // (nothing packed so far survives a refused load)
                c->err = "BULK_F16: |" + key + "| reaches " + std::to_string(w[i]) + " - outside fp16's range (6e4); load this checkpoint without the BULK_F16 option";
                return false;
            }
        return true;
    };

    // --- positional table: the buffer both modules alias; load order makes the embed_timestep key win
    const HostTensor& pe = c->sd["embed_timestep.sequence_pos_encoder.pe"];
    c->pe_len = (int)pe.shape[0];
    if (c->pe_len < c->Tq) return c->fail(RGN_ERR_BAD_SHAPE, "positional table shorter than the sequence");
    c->off_pe = blob_put(c, pe.v.data(), pe.v.size() * 4);

    // --- input stage: fold fuse_process into the two pose embeddings (concat), fp64
    {
        std::vector<double> wx, wc;
        std::vector<float> bconst(d), wxf((size_t)d * F), wcf((size_t)d * F);
        const float* win = W("input_process.poseEmbedding.weight");
        const float* wcm = W("cmo_process.poseEmbedding.weight");
        const float* bin = W("input_process.poseEmbedding.bias");
        const float* bcm = W("cmo_process.poseEmbedding.bias");
        if (c->cfg.cm_mode == RGN_CM_CONCAT) {
            const float* wf = W("fuse_process.weight");  // [d, 2d] = [Wf_x | Wf_c]
            const float* bf = W("fuse_process.bias");
            std::vector<float> wfx((size_t)d * d), wfc((size_t)d * d);
            for (int n = 0; n < d; ++n)
                for (int j = 0; j < d; ++j) {
                    wfx[(size_t)n * d + j] = wf[(size_t)n * 2 * d + j];
                    wfc[(size_t)n * d + j] = wf[(size_t)n * 2 * d + d + j];
                }
            matmul64(wfx.data(), win, d, d, F, wx);
            matmul64(wfc.data(), wcm, d, d, F, wc);
            for (int n = 0; n < d; ++n) {
                double b = bf[n];
                for (int j = 0; j < d; ++j) b += (double)wfx[(size_t)n * d + j] * bin[j] + (double)wfc[(size_t)n * d + j] * bcm[j];
                bconst[n] = (float)b;
            }
            for (size_t i = 0; i < wx.size(); ++i) {
                wxf[i] = (float)wx[i];
                wcf[i] = (float)wc[i];
            }
        } else {
            memcpy(wxf.data(), win, wxf.size() * 4);
            memcpy(wcf.data(), wcm, wcf.size() * 4);
            for (int n = 0; n < d; ++n) bconst[n] = (float)((double)bin[n] + (double)bcm[n]);
        }
        if (c->bulk_f16 && !f16_range("fuse_process.weight x input_process.poseEmbedding.weight (folded)", wxf.data(), wxf.size())) return RGN_ERR_UNSUPPORTED;
        c->lin_x = pack_linear(c, wxf.data(), nullptr, d, F, true, c->cfg.precision == RGN_PREC_BF16_X3TAIL, false, c->bulk_f16);   // fragment order: k_step
        c->lin_c = pack_linear(c, wcf.data(), bconst.data(), d, F);
    }
    c->lin_t0 = pack_linear(c, W("embed_timestep.time_embed.0.weight"), W("embed_timestep.time_embed.0.bias"), d, d);
    c->lin_t2 = pack_linear(c, W("embed_timestep.time_embed.2.weight"), W("embed_timestep.time_embed.2.bias"), d, d);

    // --- layers; cross-attention folded: G_l = Wo_c * Wv_c, g_l = Wo_c * bv_c + bo_c (1-token memory)
    std::vector<float> gall((size_t)c->L * d * d), gb((size_t)c->L * d);
    c->layers.resize(c->L);
    for (int l = 0; l < c->L; ++l) {
        const std::string p = "seqTransDecoder.layers." + std::to_string(l) + ".";
        LayerW& lw = c->layers[l];
        if (c->bulk_f16)
            for (const char* nm : {"self_attn.in_proj_weight", "self_attn.out_proj.weight", "linear1.weight", "linear2.weight"})
                if (!f16_range(p + nm, W(p + nm), c->sd[p + nm].v.size())) return RGN_ERR_UNSUPPORTED;
        lw.qkv = pack_linear(c, W(p + "self_attn.in_proj_weight"), W(p + "self_attn.in_proj_bias"), 3 * d, d, true,
                             c->cfg.precision == RGN_PREC_BF16_X3TAIL || c->cfg.precision == RGN_PREC_BF16X3,
                             c->cfg.precision == RGN_PREC_BF16_X3TAIL || c->cfg.precision == RGN_PREC_BF16X3,
                             c->bulk_f16);   // fragment order: k_rowgemm (long sequences) / k_qkv_attn_rs; + the lo plane: k_qkv_attn_rs_x3 (split phase)
        const bool fr = c->cfg.precision == RGN_PREC_BF16_X3TAIL;   // k_rowgemm operands (plain-bf16 phase)
        // (+ lo fragment planes: the operand pairs of k_mlp_x3, the split-bf16 layer tail, in every mode that has a split-bf16 phase)
        const bool frx = c->cfg.precision == RGN_PREC_BF16_X3TAIL || c->cfg.precision == RGN_PREC_BF16X3;
        lw.out = pack_linear(c, W(p + "self_attn.out_proj.weight"), W(p + "self_attn.out_proj.bias"), d, d, true, fr || frx, frx, c->bulk_f16);
        lw.ff1 = pack_linear(c, W(p + "linear1.weight"), W(p + "linear1.bias"), ff, d, true, fr || frx, frx, c->bulk_f16);
        lw.ff2 = pack_linear(c, W(p + "linear2.weight"), W(p + "linear2.bias"), d, ff, true, fr || frx, frx, c->bulk_f16);
        const char* names[6] = {"norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "norm3.weight", "norm3.bias"};
        for (int i = 0; i < 6; ++i) lw.ln[i] = blob_put(c, W(p + names[i]), (size_t)d * 4);
        const float* wv = W(p + "multihead_attn.in_proj_weight") + (size_t)2 * d * d;
        const float* bv = W(p + "multihead_attn.in_proj_bias") + 2 * d;
        const float* wo = W(p + "multihead_attn.out_proj.weight");
        const float* bo = W(p + "multihead_attn.out_proj.bias");
        std::vector<double> G;
        matmul64(wo, wv, d, d, d, G);
        for (size_t i = 0; i < G.size(); ++i) gall[(size_t)l * d * d + i] = (float)G[i];
        for (int n = 0; n < d; ++n) {
            double b = bo[n];
            for (int j = 0; j < d; ++j) b += (double)wo[(size_t)n * d + j] * bv[j];
            gb[(size_t)l * d + n] = (float)b;
        }
    }
    c->lin_g = pack_linear(c, gall.data(), gb.data(), c->L * d, d);
    if (c->bulk_f16 && !f16_range("output_process.poseFinal.weight", W("output_process.poseFinal.weight"), (size_t)F * d)) return RGN_ERR_UNSUPPORTED;
    c->lin_out = pack_linear(c, W("output_process.poseFinal.weight"), W("output_process.poseFinal.bias"), F, d, true,
                             c->cfg.precision == RGN_PREC_BF16_X3TAIL, false, c->bulk_f16);   // fragment order: k_step
    if (c->cfg.cond_mode == RGN_COND_TEXT) {
        c->lin_text = pack_linear(c, W("embed_text.weight"), W("embed_text.bias"), d, c->cfg.clip_dim);
        c->off_bt = c->lin_text.b;
    }
    if (c->cfg.cond_mode == RGN_COND_ACTION) {
        const HostTensor& a = c->sd["embed_action.action_embedding"];
        c->off_action = blob_put(c, a.v.data(), a.v.size() * 4);
    }
    c->blob_bytes = align_up(c->hblob.size(), 256);
    c->hblob.resize(c->blob_bytes);
    RGN_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->dblob), c->blob_bytes));
    RGN_HIP(c, hipMemcpy(c->dblob, c->hblob.data(), c->blob_bytes, hipMemcpyHostToDevice));
    c->hblob.clear();
    c->hblob.shrink_to_fit();
    c->sd.clear();

    // --- workspace
    const size_t B = c->cfg.max_batch, Bm = 2 * B, M = Bm * c->Tq, Mb = B * c->Tq;
    int rc;
    if ((rc = ws_alloc(c, &c->xin, Mb * F))) return rc;
    if ((rc = ws_alloc(c, &c->cmo_in, Mb * F))) return rc;
    if ((rc = ws_alloc(c, &c->c0, M * d))) return rc;
    if ((rc = ws_alloc(c, &c->h, M * d))) return rc;
    if ((rc = ws_alloc(c, &c->tmp, M * d))) return rc;
    if ((rc = ws_alloc(c, &c->qkv, M * 3 * d))) return rc;
    if ((rc = ws_alloc(c, &c->att, M * d))) return rc;
    if ((rc = ws_alloc(c, &c->ffn, M * ff))) return rc;
    if ((rc = ws_alloc(c, &c->x0tok, M * F))) return rc;
    if ((rc = ws_alloc(c, &c->pe_rows, Bm * d))) return rc;
    if ((rc = ws_alloc(c, &c->emb1, Bm * d))) return rc;
    if ((rc = ws_alloc(c, &c->emb, Bm * d))) return rc;
    if ((rc = ws_alloc(c, &c->call, Bm * c->L * d))) return rc;
    if ((rc = ws_alloc(c, &c->condemb, Bm * d))) return rc;
    if ((rc = ws_alloc(c, &c->scale, B))) return rc;
    if ((rc = ws_alloc(c, &c->te_all, (size_t)1024 * d))) return rc;
    if ((rc = ws_alloc(c, &c->sched_tmp, (size_t)2 * 1024 * d))) return rc;
    if ((rc = ws_alloc(c, &c->call_time, (size_t)1024 * c->L * d))) return rc;
    if ((rc = ws_alloc(c, &c->call_cond, Bm * c->L * d))) return rc;
    if (c->cfg.precision != RGN_PREC_F32) {
        const size_t Fp = align_up((size_t)F, 32), ffp = align_up((size_t)ff, 32);
        if ((rc = ws_alloc(c, &c->xin_hi, M * Fp))) return rc;
        if ((rc = ws_alloc(c, &c->xin_lo, M * Fp))) return rc;
        if ((rc = ws_alloc(c, &c->c0h, M * d))) return rc;
        if (c->bulk_f16 && (rc = ws_alloc(c, &c->c0h16, M * d))) return rc;
        if ((rc = ws_alloc(c, &c->h_hi, M * d))) return rc;
        if ((rc = ws_alloc(c, &c->h_lo, M * d))) return rc;
        if ((rc = ws_alloc(c, &c->att_hi, M * d))) return rc;
        if ((rc = ws_alloc(c, &c->att_lo, M * d))) return rc;
        if ((rc = ws_alloc(c, &c->ffn_hi, M * ffp))) return rc;
        if ((rc = ws_alloc(c, &c->ffn_lo, M * ffp))) return rc;
        RGN_HIP(c, hipMemset(c->xin_hi, 0, M * Fp * 2));   // K padding columns (and emb_trans_dec rows) must read as 0
        RGN_HIP(c, hipMemset(c->xin_lo, 0, M * Fp * 2));
        RGN_HIP(c, hipMemset(c->ffn_hi, 0, M * ffp * 2));
        RGN_HIP(c, hipMemset(c->ffn_lo, 0, M * ffp * 2));
        RGN_HIP(c, configure_gemm_x3());
        // measured slower than GEMM + k_layernorm at B=256 (heavy epilogue, 64-row tiles): opt-in only
        c->attn_x3 = attn_x3_supported(c->Tq, d / c->H);
        if (c->attn_x3) {
            c->Tqp = (c->Tq + 31) / 32 * 32;
            const size_t n = Bm * c->H * (size_t)c->Tqp * (d / c->H);
            __bf16** bufs[6] = {&c->q_hi, &c->q_lo, &c->k_hi, &c->k_lo, &c->vt_hi, &c->vt_lo};
            for (auto bp : bufs) {
                if ((rc = ws_alloc(c, bp, n))) return rc;
                RGN_HIP(c, hipMemset(*bp, 0, n * 2));   // padding tokens (t >= Tq) are never written and must read as 0
            }
            RGN_HIP(c, configure_attn_x3(c->Tq, d / c->H));
        }
        c->fuse_qkv = qkv_attn_supported(c->Tq, d / c->H, d) && !opt_flag(c, "NO_FUSED_QKV");
        { int v; if (opt_get(c, "BIG_TILE_ROWS", &v)) c->big_tile_rows = v; }
        c->rowgemm = c->cfg.precision == RGN_PREC_BF16_X3TAIL && !opt_flag(c, "NO_ROWGEMM") && c->Tq >= 8 &&   // (8 rows of a wave: <= 2 samples)
                     rowgemm_supported(d, d, true) && rowgemm_supported(d, (int)align_up((size_t)ff, 32), true) &&
                     rowgemm_supported(ff, d, false);
        if (c->rowgemm) RGN_HIP(c, configure_rowgemm());
        c->mlp = c->rowgemm && mlp_supported(d, ff, c->Tq) && !opt_flag(c, "NO_MLP");
        if (c->mlp) RGN_HIP(c, configure_mlp());
        {   // the split-bf16 layer tail as one kernel (REGENNET_MLP_X3=0: k_gemm_x3 x 3 + k_layernorm x 2 per layer instead)
            int v = 1;
            (void)opt_get(c, "MLP_X3", &v);
            c->mlp_x3 = v != 0 && mlp_x3_supported(d, ff, c->Tq);
            if (c->mlp_x3) RGN_HIP(c, configure_mlp_x3());
        }
        if (c->fuse_qkv) RGN_HIP(c, configure_qkv_attn());
        c->qkv_rs = !opt_flag(c, "NO_QKV_RS");
        c->qkv_x3_dma = opt_flag(c, "QKV_X3_DMA");
        c->step_fused = c->rowgemm && !c->etd && c->lin_x.fr && c->lin_out.fr && c->lin_out.has_bias && !c->lin_x.has_bias &&
                        step_fused_supported(d, F, c->lin_x.Kp) && !opt_flag(c, "NO_STEP_FUSION");
        if (c->step_fused) RGN_HIP(c, configure_step());
        // one workgroup per sample costs a full 64-row tile whatever the length, the kernel-per-stage chain costs the rows there are, and the
        // fused form is worth ~20 % of a layer: it takes evaluations of at least 52 tokens per sample (REGENNET_LAYERS_MIN_TQ overrides: tests)
        int ly_min_tq = 52, ly_on = 1, ly_steps = 1;
        (void)opt_get(c, "LAYERS_MIN_TQ", &ly_min_tq);
        (void)opt_get(c, "LAYERS", &ly_on);
        (void)opt_get(c, "LAYERS_STEPS", &ly_steps);
        c->layers_fused = c->mlp && c->fuse_qkv && c->qkv_rs && layers_supported(d, ff, c->H, c->Tq, c->L) && c->Tq >= ly_min_tq &&
                          ly_on != 0;
        if (c->layers_fused) RGN_HIP(c, configure_layers());
        { int v; if (opt_get(c, "LAYERS_MIN_B", &v)) c->layers_min_b = c->layers_min_b_default = v < 1 ? 1 : v; }
        { int v; if (opt_get(c, "LAYERS_GUIDED", &v)) c->layers_guided = c->layers_guided_default = v < 0 ? 1 : (v > 2 ? 2 : v); }
        { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->cfg.device) == hipSuccess && v > 0) c->num_cus = v; }
        c->layers_steps = c->layers_fused && c->step_fused && layers_steps_supported(d, F, c->lin_x.Kp) &&
                          ly_steps != 0;
        c->step_no_quads = opt_flag(c, "STEP_NO_QUADS");
        c->qkv_long = c->cfg.precision == RGN_PREC_BF16_X3TAIL && qkv_attn_long_supported(c->Tq, d / c->H, d) && !opt_flag(c, "NO_QKV_LONG");
        if (c->qkv_long) RGN_HIP(c, configure_qkv_attn_long());
        // the forms with fp16 instantiations: the multi-step one-kernel stack; k_qkv_attn_long + k_mlp2 + k_step (prec_plan decides per batch)
        c->bulk_f16 = c->bulk_f16 && c->step_fused && (c->layers_steps || ((c->qkv_long || (c->fuse_qkv && c->qkv_rs && d == 512)) && c->mlp));
        c->sb = c->attn_x3 && sb_supported(d, ff, d / c->H);
        { int v; if (opt_get(c, "SB_FUSED_ATTN", &v)) c->sb_attn = v != 0; }
        { int v; if (opt_get(c, "SB_ROWS", &v)) c->sb_rows = c->sb_rows_default = v < 0 ? 0 : v; }
        if (c->sb) RGN_HIP(c, configure_sb());
        if (c->sb) RGN_HIP(c, configure_sb_qkv_attn());
    }
    if ((rc = ws_alloc(c, &c->d_tab, (size_t)1024))) return rc;
    if ((rc = ws_alloc(c, &c->d_step, (size_t)4 + 1 + B))) return rc;   // [0] loop index, [3] scratch, [4 ..] k_update's ticket counters
    if ((rc = ws_alloc(c, &c->d_sp, (size_t)1))) return rc;
    RGN_HIP(c, configure_attention(c->Tq, c->d / c->H));
    RGN_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    RGN_HIP(c, hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
    RGN_HIP(c, hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
    for (int i = 0; i < rgn_ctx::MAX_SIDE; ++i) {
        RGN_HIP(c, hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
        RGN_HIP(c, hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    }
    RGN_HIP(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    { int v; if (opt_get(c, "BULK_RESID_LO", &v)) c->bulk_resid_lo = v != 0; }
    { int v; if (opt_get(c, "GRAPH_STEPS", &v)) c->graph_steps = v < 1 ? 1 : (v > 100 ? 100 : v); }
    int v_streams;
    if (opt_get(c, "STREAMS", &v_streams)) {
        c->nchains = v_streams < 1 ? 1 : (v_streams > 16 ? 16 : v_streams);
        c->nchains_user = true;
    }
    RGN_HIP(c, hipMemset(c->xin, 0, Mb * F * sizeof(float)));
    RGN_HIP(c, hipMemset(c->cmo_in, 0, Mb * F * sizeof(float)));
    RGN_HIP(c, hipMemset(c->d_step, 0, (4 + 1 + B) * sizeof(int)));
    c->finalized = true;
    return RGN_OK;
}

}  // namespace rgnh
