// Host side of libregennet_hip.so: the C-ABI declared in include/regennet_hip.h.
// Owns the packed weight blob, the activation workspace, the per-step coefficient tables, the
// step orchestration (one denoiser evaluation + sampler update: 53 kernel launches per chain of samples at L = 8) and
// its hipGraph capture. All arithmetic on tensors happens in the .hip files next to this one (rgn_gemm_x3: split-bf16
// GEMM, rgn_qkv_attn: fused in_proj + attention, rgn_attn_x3: attention for long sequences, rgn_kernels: the rest).
#include "../../include/regennet_hip.h"
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <vector>

using namespace rgn;

namespace {

thread_local std::string g_create_error;   // text of the calling thread's last failed rgn_create

struct HostTensor {
    std::vector<float> v;
    std::vector<int64_t> shape;
};

struct Lin {  // one packed nn.Linear: offsets (bytes) into the weight blob
    size_t w = 0, hi = 0, lo = 0, b = 0;   // fp32 [N,Kp]; bf16 hi/lo planes (row-major [N,Kp] or K32-blocked)
    size_t fr = 0;                         // bf16 hi plane in MFMA-fragment order (operand of k_rowgemm), optional
    size_t fr_lo = 0;                      // bf16 lo plane in the same order (operand of k_mlp_x3), optional
    size_t fr16 = 0;                       // IEEE fp16 plane in the same order (operand of k_layers<.., F16>: rgn_set_option "BULK_F16"), optional
    int N = 0, K = 0, Kp = 0;
    bool has_bias = false;
    bool blocked = false;                  // hi/lo are K32-blocked [Kp/32][N][32] (operands of k_gemm_x3)
};

struct LayerW {
    Lin qkv, out, ff1, ff2;
    size_t ln[6];  // g1,b1,g2,b2,g3,b3
};

struct ProfEv {
    int kc;
    hipEvent_t a, b;
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// round-to-nearest-even fp32 -> bf16 bits
inline uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
// round-to-nearest-even fp32 -> IEEE fp16 bits (subnormals kept, overflow -> inf: the caller has checked the range)
inline uint16_t f2h(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((a > 0x7f800000u) ? 0x200u : 0u));
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);            // >= 65520 rounds to inf
    if (a < 0x33000001u) return sign;                                   // <= 2^-25: rounds to zero
    int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;                           // 24-bit significand
    int shift = e < -14 ? (13 + (-14 - e)) : 13;                        // bits dropped (subnormal: more)
    const uint32_t halfway = 1u << (shift - 1), rem = m & ((1u << shift) - 1);
    uint32_t q = m >> shift;
    if (rem > halfway || (rem == halfway && (q & 1u))) ++q;
    const uint32_t bits = e < -14 ? q : (uint32_t)((e + 15 - 1) << 10) + q;   // (q carries the implicit one: + (e + 14) << 10)
    return (uint16_t)(sign | bits);
}
inline float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

}  // namespace

struct rgn_ctx {
    rgn_config cfg{};
    std::string err;
    std::map<std::string, std::vector<int64_t>> expected;  // key -> shape (pe: shape[0] free)
    std::map<std::string, HostTensor> sd;
    std::map<std::string, int> opts;   // rgn_set_option: per-handle kernel-selection switches (they take precedence over REGENNET_<KEY> in the environment)
    bool finalized = false, have_sched = false, have_cond = false;
    int F = 0, d = 0, Tq = 0, etd = 0, L = 0, H = 0, ff = 0, pe_len = 0;

    // packed weights
    std::vector<char> hblob;
    char* dblob = nullptr;
    size_t blob_bytes = 0;
    Lin lin_x, lin_c, lin_t0, lin_t2, lin_g, lin_out, lin_text;
    std::vector<LayerW> layers;
    size_t off_pe = 0, off_action = 0, off_bt = 0;

    // workspace
    float *xin = nullptr, *cmo_in = nullptr, *c0 = nullptr, *h = nullptr, *tmp = nullptr, *qkv = nullptr, *att = nullptr,
          *ffn = nullptr, *x0tok = nullptr, *pe_rows = nullptr, *emb1 = nullptr, *emb = nullptr, *call = nullptr,
          *condemb = nullptr, *scale = nullptr, *te_all = nullptr, *call_time = nullptr, *call_cond = nullptr, *sched_tmp = nullptr;
    __bf16* c0h = nullptr;             // bf16 copy of c0 for the fused step boundary (k_step)
    _Float16* c0h16 = nullptr;         // fp16 copy of c0 (bulk_f16)
    bool bulk_f16 = false;             // plain phase of the precision schedule on fp16 operands where k_layers<true> runs it (rgn_set_option "BULK_F16" / REGENNET_BULK_F16)
    __bf16 *xin_hi = nullptr, *xin_lo = nullptr, *h_hi = nullptr, *h_lo = nullptr, *att_hi = nullptr, *att_lo = nullptr,
           *ffn_hi = nullptr, *ffn_lo = nullptr;   // K32-blocked split planes (bf16 precision modes)
    __bf16 *q_hi = nullptr, *q_lo = nullptr, *k_hi = nullptr, *k_lo = nullptr, *vt_hi = nullptr, *vt_lo = nullptr;   // attention-ready planes
    bool attn_x3 = false;
    int Tqp = 0;
    bool fuse_qkv = false;             // in_proj GEMM + attention in one per-sample kernel (k_qkv_attn)
    int big_tile_rows = 7000;          // launches of at least this many rows per chain use the 256x256 GEMM tile
    bool rowgemm = false;              // plain-bf16 phase: row-complete GEMMs with fused LayerNorm / GELU (k_rowgemm)
    bool mlp = false;                  // plain-bf16 phase: the whole layer tail in one row-persistent kernel (k_mlp)
    bool mlp_x3 = false;               // split-bf16 phase: the whole layer tail in one row-persistent kernel (k_mlp_x3, REGENNET_MLP_X3)
    bool step_fused = false;           // plain-bf16 phase: output projection (+ guidance) + sampler update + next input embedding in one kernel (REGENNET_NO_STEP_FUSION=1: three launches)
    bool layers_fused = false;         // plain-bf16 phase, <= 64 tokens: the whole decoder stack of an evaluation as one kernel, one sample per workgroup (k_layers; REGENNET_LAYERS=0: kernel per stage)
    int layers_min_b_default = 64;     // (REGENNET_LAYERS_MIN_B)
    int layers_min_b = 64;             // ... for evaluations of at least this many samples (REGENNET_LAYERS_MIN_B): one workgroup per sample is a latency chain of 8 layers (250-step calls: 114 ms at any B <= 256), the kernel-per-stage form spreads a sample over more CUs (B = 16 / 32 / 48: 110-111 ms; B = 64: 114.4 vs 113.6)
    bool layers_steps = false;         // ... and, unguided, whole runs of sampler steps in ONE launch (k_layers<true>: stack + step boundary per sample; REGENNET_LAYERS_STEPS=0: one k_layers + one k_step per step)
    bool layers_guided = true;         // ... and guided runs too (a motion's two evaluations in one workgroup; REGENNET_LAYERS_GUIDED=0: k_layers per evaluation + the guided k_step)
    bool skip_embed_out = false;       // (set by run_eval around run_layers while it enqueues a fused step)
    int step_no_quads = 0;             // REGENNET_STEP_NO_QUADS=1 (tests)
    bool qkv_rs = true;                // plain-bf16 phase: k_qkv_attn with register-streamed weights (REGENNET_NO_QKV_RS=1: the DMA-fed loop)
    bool qkv_long = false;             // plain-bf16 phase, 65 .. 160 tokens: fused in_proj + attention per (sample, head) (REGENNET_NO_QKV_LONG=1: in_proj GEMM + k_attn_x3)
    bool sb = false;                   // small-batch engine: column-split GEMMs with consumer-side LayerNorm (k_sb_gemm)
    bool sb_attn = false;              // small-batch engine: in_proj + attention as one launch per layer (k_sb_qkv_attn; REGENNET_SB_FUSED_ATTN=1: it loses below B ~ 6)
    int sb_rows = 640;                 // evaluations of at most this many token rows take it (rgn_set_small_batch_rows; 0 disables)
    int sb_rows_default = 640;         // (REGENNET_SB_ROWS): measured crossover with the throughput kernels at 60 tokens: between B = 10 and 11
                                       // (round 3, 250-step calls: B = 10 108 vs 116 ms, B = 11 122 vs 117, B = 12 123 vs 117; it was B = 12 .. 16 in round 2)
    StepCoef* d_tab = nullptr;
    int* d_step = nullptr;
    SampleParams* d_sp = nullptr;
    std::vector<void*> allocs;
    hipStream_t stream = nullptr;      // all work runs here; callers' streams are joined by events
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    static constexpr int MAX_SIDE = 15;
    hipStream_t side[MAX_SIDE] = {};   // extra chains of the multi-stream evaluation
    hipEvent_t ev_fork = nullptr, ev_join[MAX_SIDE] = {};
    int nchains = 4;                   // REGENNET_STREAMS = 1 .. 16 (default 4; 2 for evaluations of 129 .. 256 row tiles, see run_eval)
    bool nchains_user = false;         // REGENNET_STREAMS was given: no size rule
    // Precision schedule (RGN_PREC_BF16_X3TAIL): the loop indices i >= x3_tail run plain-bf16 GEMMs (one MFMA per
    // product, hi planes only as GEMM operands), the last x3_tail indices and every rgn_denoise call the split-bf16 ones.
    // phase_x3 is the phase of the evaluation being enqueued / captured.
    bool phase_x3 = true;
    bool phase_f16 = false;            // ... and, for a plain evaluation: fp16 operands (the fp16 sub-phase of the schedule, rgn_set_f16_steps)
    int x3_tail = -1;                  // -1: default_tail(S)
    int f16_steps = -1;                // rgn_set_f16_steps: plain-phase steps right in front of the split-bf16 tail that run on fp16 operands (-1: default)
    int const_noise = 0;               // rgn_set_const_noise
    bool bulk_resid_lo = false;        // bulk phase: residual stream as the hi plane only (REGENNET_BULK_RESID_LO=1: hi + lo; the switch-point
                                       // sweeps measure the same final error either way, hi-only is ~6 % faster)

    // schedule (host copies)
    int S = 0;
    std::vector<int64_t> tmap;
    std::vector<double> coef1, coef2, logvar, srecip, srecipm1, ac, acp;
    float tab_eta = -1.f;
    bool tab_valid = false;

    // bound condition
    int B = 0;
    int xin_rows = -1;                 // row count the xin planes are currently laid out for
    bool cond_has_scale = false;

    // graphs: key = B | guided<<20 | sampler<<21 | phase_x3<<23 | steps<<24
    int graph_steps = 10;              // loop iterations per captured graph for long ranges (REGENNET_GRAPH_STEPS)
    std::map<uint64_t, hipGraphExec_t> graphs;

    // profiling
    bool prof = false;
    std::vector<ProfEv> prof_pool;     // pre-created event pairs
    size_t prof_used = 0;
    double prof_ms[KC_COUNT] = {0};
    double prof_bracket_ms = -1.0;     // event-pair time around a no-op kernel (calibrated on first enable)
    int64_t prof_n[KC_COUNT] = {0};

    int fail(int code, const std::string& m) {
        err = m;
        return code;
    }
    template <typename T>
    T* dp(size_t off) const { return reinterpret_cast<T*>(dblob + off); }
};

namespace {

const char* kclass_names[KC_COUNT] = {"gemm_mfma", "attention", "layernorm", "embed", "update", "misc", "qkv_attn", "rowgemm_ln", "rowgemm_act", "mlp", "sb_gemm", "step_fused", "layers", "steps_fused"};

#define RGN_HIP(h, expr)                                                                                    \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess)                                                                               \
            return (h)->fail(RGN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));               \
    } while (0)

// Launch wrapper: optional HIP-event bracketing per kernel class (eager mode only).
// Launch wrapper: optional HIP-event bracketing per kernel class. Events come from a pool created by
// rgn_profile_enable (no creation cost between launches); launches beyond the pool are simply not timed.
#define RGN_LAUNCH(h, KCLS, stream, call)                                        \
    do {                                                                         \
        const bool _p = (h)->prof && (h)->prof_used < (h)->prof_pool.size();     \
        if (_p) {                                                                \
            (h)->prof_pool[(h)->prof_used].kc = (KCLS);                          \
            (void)hipEventRecord((h)->prof_pool[(h)->prof_used].a, (stream));    \
        }                                                                        \
        RGN_HIP(h, call);                                                        \
        if (_p) {                                                                \
            (void)hipEventRecord((h)->prof_pool[(h)->prof_used].b, (stream));    \
            (h)->prof_used++;                                                    \
        }                                                                        \
    } while (0)

// Work is enqueued on the handle's own non-blocking stream (capturable, unlike the legacy null stream
// torch hands over by default) and ordered after / before the caller's stream with events.
int stream_enter(rgn_ctx* c, hipStream_t user) {
    RGN_HIP(c, hipEventRecord(c->ev_in, user));
    RGN_HIP(c, hipStreamWaitEvent(c->stream, c->ev_in, 0));
    return RGN_OK;
}
int stream_exit(rgn_ctx* c, hipStream_t user) {
    RGN_HIP(c, hipEventRecord(c->ev_out, c->stream));
    RGN_HIP(c, hipStreamWaitEvent(user, c->ev_out, 0));
    return RGN_OK;
}

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
size_t blob_put(rgn_ctx* c, const void* src, size_t bytes) {
    const size_t off = align_up(c->hblob.size(), 256);
    c->hblob.resize(off + bytes);
    if (src) memcpy(c->hblob.data() + off, src, bytes);
    return off;
}

// Pack W[N,K] (row-major fp32) into fp32 [N,Kp] plus bf16 hi/lo planes; bias optional.
Lin pack_linear(rgn_ctx* c, const float* W, const float* bias, int N, int K, bool blocked = false, bool frag = false, bool frag_lo = false, bool frag16 = false) {
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
void matmul64(const float* A, const float* Bm, int N, int J, int K, std::vector<double>& C) {
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
int ws_alloc(rgn_ctx* c, T** p, size_t count) {
    void* q = nullptr;
    RGN_HIP(c, hipMalloc(&q, count * sizeof(T) + 256));
    c->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return RGN_OK;
}

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

// precision of the per-schedule / per-condition / rgn_denoise-only GEMMs (k_gemm_f32 / k_gemm_bf16): the schedule mode
// runs them split-bf16 (they are once-per-call work)
inline int small_prec(const rgn_ctx* c) { return c->cfg.precision == RGN_PREC_BF16_X3TAIL ? RGN_PREC_BF16X3 : c->cfg.precision; }
// split-bf16 (three MFMAs per product) for the evaluation being enqueued?
inline bool eval_x3(const rgn_ctx* c) {
    return c->cfg.precision == RGN_PREC_BF16X3 || (c->cfg.precision == RGN_PREC_BF16_X3TAIL && c->phase_x3);
}
// do activation planes carry a lo part at all (allocation, sampler state, residual stream)?
inline bool has_lo(const rgn_ctx* c) { return c->cfg.precision == RGN_PREC_BF16X3 || c->cfg.precision == RGN_PREC_BF16_X3TAIL; }
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

// Small-batch evaluation (rgn_sb.hip): the same embedding GEMM + L decoder layers + output projection as run_layers for ALL
// rows of the evaluation on one stream, as column-split kernels: 5 launches per layer, pre-norm sums in `tmp` (fp32), the
// residual stream in `h` (fp32), LayerNorms applied by the consuming GEMM.
bool use_sb(const rgn_ctx* c, int rows) { return c->sb && c->cfg.precision != RGN_PREC_F32 && rows <= c->sb_rows; }

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
    p.steps = p.layers && p.step_fused && c->layers_steps && (!guided || (c->layers_guided && c->ffn_hi &&
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
struct PrecPlan { int tail = 0, n16 = 0; };
PrecPlan prec_plan(const rgn_ctx* c, const Dims& dm, bool guided) {
    PrecPlan pp;
    if (c->cfg.precision != RGN_PREC_BF16_X3TAIL) return pp;
    const EvalPlan plain = plan_eval(c, dm, guided, false, true);
    // (the forms with an fp16 instantiation: the multi-step one-kernel stack, and the kernel-per-stage chain of 150-frame models -
    //  k_qkv_attn_long + k_mlp2 + k_step, whose planes hand the residual stream from step to step)
    const bool f16_ok = c->bulk_f16 && (plain.steps || (plain.step_fused && !plain.layers && plain.attn == AF_QKV_LONG && plain.tail == TF_MLP));
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
        fill_layers_args(c, g, dm, sampling, ccond_rows, s0);
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
            g.Wfr = (w.qkv.fr && c->qkv_rs) ? c->dp<__bf16>(w.qkv.fr) : nullptr;   // plain-bf16 phase: weights streamed to registers
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

// No C++ exception crosses the C boundary (include/regennet_hip.h): every entry point runs its body inside this guard, and what the
// host-side containers may throw (std::bad_alloc, std::length_error ...) comes back as RGN_ERR_INTERNAL with the text in rgn_last_error.
int boundary_error(rgn_ctx* h, const char* fn, const char* what) noexcept {
    try {
        std::string m = std::string(fn) + ": C++ exception at the boundary: " + what;
        if (h) h->err.swap(m);
        else g_create_error.swap(m);
    } catch (...) {   // (not even the message could be built: the code alone reports it)
    }
    return RGN_ERR_INTERNAL;
}
template <class F>
int rgn_guard(rgn_ctx* h, const char* fn, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return boundary_error(h, fn, "std::bad_alloc (host memory)");
    } catch (const std::exception& e) {
        return boundary_error(h, fn, e.what());
    } catch (...) {
        return boundary_error(h, fn, "unknown exception");
    }
}

}  // namespace

extern "C" {

const char* rgn_last_error(rgn_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int rgn_create(const rgn_config* cfg, rgn_handle* out) {
    return rgn_guard(static_cast<rgn_ctx*>(nullptr), "rgn_create", [&]() -> int {
        if (!cfg || !out) {
            g_create_error = "rgn_create: null argument";
            return RGN_ERR_INVALID_ARG;
        }
        *out = nullptr;
        auto bad = [&](int code, const std::string& m) {
            g_create_error = m;
            return code;
        };
        if (cfg->njoints <= 0 || cfg->nfeats <= 0 || cfg->num_frames <= 0 || cfg->latent_dim <= 0 || cfg->ff_size <= 0 ||
            cfg->num_heads <= 0 || cfg->num_layers <= 0 || cfg->max_batch <= 0)
            return bad(RGN_ERR_INVALID_ARG, "rgn_create: non-positive dimension");
        if (cfg->latent_dim % cfg->num_heads) return bad(RGN_ERR_INVALID_ARG, "rgn_create: latent_dim % num_heads != 0");
        if (cfg->latent_dim % 64 || cfg->latent_dim > 1024 || (cfg->latent_dim / 64 & (cfg->latent_dim / 64 - 1)))
            return bad(RGN_ERR_UNSUPPORTED, "rgn_create: latent_dim must be 64*2^k <= 1024");
        if (cfg->num_frames > 4096) return bad(RGN_ERR_UNSUPPORTED, "rgn_create: more than 4096 frames (Philox element counter)");
        if (cfg->latent_dim / cfg->num_heads > 128)
            return bad(RGN_ERR_UNSUPPORTED, "rgn_create: head dim > 128 unsupported");
        if (cfg->cm_mode != RGN_CM_ADD && cfg->cm_mode != RGN_CM_CONCAT) return bad(RGN_ERR_INVALID_ARG, "rgn_create: cm_mode");
        if (cfg->cond_mode < RGN_COND_NONE || cfg->cond_mode > RGN_COND_TEXT) return bad(RGN_ERR_INVALID_ARG, "rgn_create: cond_mode");
        if (cfg->precision < RGN_PREC_F32 || cfg->precision > RGN_PREC_BF16_X3TAIL) return bad(RGN_ERR_INVALID_ARG, "rgn_create: precision");
        if (cfg->cond_mode == RGN_COND_ACTION && cfg->num_actions <= 0) return bad(RGN_ERR_INVALID_ARG, "rgn_create: num_actions");
        if (cfg->cond_mode == RGN_COND_TEXT && cfg->clip_dim <= 0) return bad(RGN_ERR_INVALID_ARG, "rgn_create: clip_dim");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return bad(RGN_ERR_HIP, "rgn_create: no HIP device visible");
        if (cfg->device < 0 || cfg->device >= ndev) return bad(RGN_ERR_INVALID_ARG, "rgn_create: device ordinal out of range");
        if (hipSetDevice(cfg->device) != hipSuccess) return bad(RGN_ERR_HIP, "rgn_create: hipSetDevice failed");

        std::unique_ptr<rgn_ctx> c(new rgn_ctx());
        c->cfg = *cfg;
        c->F = cfg->njoints * cfg->nfeats;
        c->d = cfg->latent_dim;
        c->etd = cfg->emb_trans_dec ? 1 : 0;
        c->Tq = cfg->num_frames + c->etd;
        c->L = cfg->num_layers;
        c->H = cfg->num_heads;
        c->ff = cfg->ff_size;
        build_expected(c.get());
        *out = c.release();
        return RGN_OK;
    });
}

int rgn_destroy(rgn_handle h) {
    return rgn_guard(h, "rgn_destroy", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        (void)hipSetDevice(h->cfg.device);
        (void)hipDeviceSynchronize();
        for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
        for (auto& e : h->prof_pool) {
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
        if (h->ev_in) (void)hipEventDestroy(h->ev_in);
        if (h->ev_out) (void)hipEventDestroy(h->ev_out);
        if (h->stream) (void)hipStreamDestroy(h->stream);
        for (int i = 0; i < rgn_ctx::MAX_SIDE; ++i) {
            if (h->side[i]) (void)hipStreamDestroy(h->side[i]);
            if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
        }
        if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
        for (void* p : h->allocs) (void)hipFree(p);
        if (h->dblob) (void)hipFree(h->dblob);
        delete h;
        return RGN_OK;
    });
}

int rgn_load_weight(rgn_handle h, const char* key, const float* host, const int64_t* shape, int32_t ndim) {
    return rgn_guard(h, "rgn_load_weight", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        if (!key || !host || !shape || ndim <= 0) return h->fail(RGN_ERR_INVALID_ARG, "rgn_load_weight: null/empty argument");
        if (h->finalized) return h->fail(RGN_ERR_STATE, "rgn_load_weight: weights already finalized");
        const std::string k(key);
        if (k.rfind("clip_model.", 0) == 0) return RGN_OK;  // accepted and ignored (model_util.py:8)
        auto it = h->expected.find(k);
        if (it == h->expected.end()) return h->fail(RGN_ERR_BAD_KEY, "unexpected key in state_dict: " + k);
        const auto& es = it->second;
        bool ok = (int)es.size() == ndim;
        for (int i = 0; ok && i < ndim; ++i) ok = (es[i] == -1) ? (shape[i] > 0) : (es[i] == shape[i]);
        if (!ok) {
            std::string m = "size mismatch for " + k + ": got [";
            for (int i = 0; i < ndim; ++i) m += std::to_string(shape[i]) + (i + 1 < ndim ? "," : "");
            m += "], expected [";
            for (size_t i = 0; i < es.size(); ++i) m += std::to_string(es[i]) + (i + 1 < es.size() ? "," : "");
            return h->fail(RGN_ERR_BAD_SHAPE, m + "]");
        }
        size_t n = 1;
        for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
        HostTensor t;
        t.v.assign(host, host + n);
        t.shape.assign(shape, shape + ndim);
        h->sd[k] = std::move(t);
        return RGN_OK;
    });
}

int rgn_finalize_weights(rgn_handle h) {
    return rgn_guard(h, "rgn_finalize_weights", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        if (h->finalized) return h->fail(RGN_ERR_STATE, "rgn_finalize_weights: already finalized");
        rgn_ctx* c = h;
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
        { int v; if (opt_get(c, "F16_STEPS", &v)) c->f16_steps = v < 0 ? -1 : v; }
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
                                 c->cfg.precision == RGN_PREC_BF16_X3TAIL, false, c->bulk_f16);   // fragment order: k_rowgemm (long sequences) / k_qkv_attn_rs
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
            { int v; if (opt_get(c, "LAYERS_GUIDED", &v)) c->layers_guided = v != 0; }
            c->layers_steps = c->layers_fused && c->step_fused && layers_steps_supported(d, F, c->lin_x.Kp) &&
                              ly_steps != 0;
            c->step_no_quads = opt_flag(c, "STEP_NO_QUADS");
            c->qkv_long = c->cfg.precision == RGN_PREC_BF16_X3TAIL && qkv_attn_long_supported(c->Tq, d / c->H, d) && !opt_flag(c, "NO_QKV_LONG");
            if (c->qkv_long) RGN_HIP(c, configure_qkv_attn_long());
            // the forms with fp16 instantiations: the multi-step one-kernel stack; k_qkv_attn_long + k_mlp2 + k_step (prec_plan decides per batch)
            c->bulk_f16 = c->bulk_f16 && c->step_fused && (c->layers_steps || (c->qkv_long && c->mlp));
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
    });
}

int rgn_weight_blob(rgn_handle h, void** dev_ptr, uint64_t* nbytes) {
    return rgn_guard(h, "rgn_weight_blob", [&]() -> int {
        if (!h || !dev_ptr || !nbytes) return RGN_ERR_INVALID_ARG;
        if (!h->finalized) return h->fail(RGN_ERR_STATE, "rgn_weight_blob: weights not finalized");
        *dev_ptr = h->dblob;
        *nbytes = h->blob_bytes;
        return RGN_OK;
    });
}

int rgn_set_schedule(rgn_handle h, const rgn_schedule* s) {
    return rgn_guard(h, "rgn_set_schedule", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        if (!s || s->S <= 0 || !s->timestep_map || !s->posterior_mean_coef1 || !s->posterior_mean_coef2 || !s->model_log_variance ||
            !s->sqrt_recip_alphas_cumprod || !s->sqrt_recipm1_alphas_cumprod || !s->alphas_cumprod || !s->alphas_cumprod_prev)
            return h->fail(RGN_ERR_INVALID_ARG, "rgn_set_schedule: null table or S <= 0");
        if (!h->finalized) return h->fail(RGN_ERR_STATE, "rgn_set_schedule: weights not finalized");
        if (s->S > 1024) return h->fail(RGN_ERR_UNSUPPORTED, "rgn_set_schedule: more than 1024 steps");
        for (int i = 0; i < s->S; ++i) {
            if (s->timestep_map[i] < 0 || s->timestep_map[i] >= h->pe_len)
                return h->fail(RGN_ERR_INVALID_ARG, "rgn_set_schedule: timestep_map entry outside the positional table");
            if (i && s->timestep_map[i] <= s->timestep_map[i - 1])
                return h->fail(RGN_ERR_INVALID_ARG, "rgn_set_schedule: timestep_map must be strictly increasing");
        }
        h->S = s->S;
        h->tmap.assign(s->timestep_map, s->timestep_map + s->S);
        h->coef1.assign(s->posterior_mean_coef1, s->posterior_mean_coef1 + s->S);
        h->coef2.assign(s->posterior_mean_coef2, s->posterior_mean_coef2 + s->S);
        h->logvar.assign(s->model_log_variance, s->model_log_variance + s->S);
        h->srecip.assign(s->sqrt_recip_alphas_cumprod, s->sqrt_recip_alphas_cumprod + s->S);
        h->srecipm1.assign(s->sqrt_recipm1_alphas_cumprod, s->sqrt_recipm1_alphas_cumprod + s->S);
        h->ac.assign(s->alphas_cumprod, s->alphas_cumprod + s->S);
        h->acp.assign(s->alphas_cumprod_prev, s->alphas_cumprod_prev + s->S);
        h->tab_valid = false;
        h->have_sched = true;
        RGN_HIP(h, hipSetDevice(h->cfg.device));
        int rc = build_step_table(h, 0.0f);
        if (rc) return rc;
        // per-step timestep embedding TE[i] = time_embed(pe[timestep_map[i]]) and its folded cross-attention image
        // call_time[i] = TE[i] . G^T + g  (cmdm.py:297-298 + the 1-token multihead_attn of every layer), once per schedule
        rgn_ctx* c = h;
        hipStream_t es = c->stream;
        const int d = c->d, S = c->S;
        RGN_LAUNCH(c, KC_EMBED, es, launch_gather_pe_all(c->dp<float>(c->off_pe), c->d_tab, c->sched_tmp, S, d, es));
        GemmArgs g = gemm_args(c, c->lin_t0, c->sched_tmp, d, c->sched_tmp + (size_t)1024 * d, d, S);
        g.act = 2;
        RGN_LAUNCH(c, KC_GEMM, es, launch_gemm(g, small_prec(c), es));
        g = gemm_args(c, c->lin_t2, c->sched_tmp + (size_t)1024 * d, d, c->te_all, d, S);
        RGN_LAUNCH(c, KC_GEMM, es, launch_gemm(g, small_prec(c), es));
        g = gemm_args(c, c->lin_g, c->te_all, d, c->call_time, c->L * d, S);
        RGN_LAUNCH(c, KC_GEMM, es, launch_gemm(g, small_prec(c), es));
        RGN_HIP(c, hipStreamSynchronize(es));
        return RGN_OK;
    });
}

int rgn_set_condition(rgn_handle h, int32_t B, const float* cmotion, const int64_t* action, const float* text_feat,
                      const float* scale, void* stream) {
    return rgn_guard(h, "rgn_set_condition", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        rgn_ctx* c = h;
        if (!c->finalized) return c->fail(RGN_ERR_STATE, "rgn_set_condition: weights not finalized");
        if (B <= 0 || B > c->cfg.max_batch) return c->fail(RGN_ERR_INVALID_ARG, "rgn_set_condition: B outside (0, max_batch]");
        if (!cmotion) return c->fail(RGN_ERR_INVALID_ARG, "rgn_set_condition: y['cmotion'] is required (cmdm.py:189)");
        if (c->cfg.cond_mode == RGN_COND_ACTION && !action) return c->fail(RGN_ERR_INVALID_ARG, "rgn_set_condition: y['action'] required");
        if (c->cfg.cond_mode == RGN_COND_TEXT && !text_feat) return c->fail(RGN_ERR_INVALID_ARG, "rgn_set_condition: text features required");
        hipStream_t us = reinterpret_cast<hipStream_t>(stream), s = c->stream;
        RGN_HIP(c, hipSetDevice(c->cfg.device));
        int rc0 = stream_enter(c, us);
        if (rc0) return rc0;
        const Dims dm = make_dims(c, B, false);
        const int d = c->d;
        // hoisted: c0 = cmo_process(cmotion) -> fuse half + all constant biases + positional encoding
        RGN_LAUNCH(c, KC_UPDATE, s, launch_pack_x(cmotion, c->cmo_in, Planes{nullptr, nullptr, 0}, 1, dm, s));
        GemmArgs g = gemm_args(c, c->lin_c, c->cmo_in, c->F, c->c0, d, B * dm.Tq);
        RGN_LAUNCH(c, KC_GEMM, s, launch_gemm(g, small_prec(c), s));
        if (!c->cfg.wo_pos_emb) RGN_LAUNCH(c, KC_EMBED, s, launch_add_pe(c->c0, c->dp<float>(c->off_pe), dm, s));
        RGN_HIP(c, hipMemcpyAsync(c->c0 + (size_t)B * dm.Tq * d, c->c0, (size_t)B * dm.Tq * d * sizeof(float), hipMemcpyDeviceToDevice, s));   // uncond half
        if (c->c0h) RGN_LAUNCH(c, KC_EMBED, s, launch_cvt_bf16(c->c0, c->c0h, (size_t)2 * B * dm.Tq * d, s));   // k_step's copy (plain-bf16 phase only)
        if (c->c0h16) RGN_LAUNCH(c, KC_EMBED, s, launch_cvt_f16(c->c0, c->c0h16, (size_t)2 * B * dm.Tq * d, s));  // ... and the fp16-operand form's
        // condition embedding rows: [0,B) conditional, [B,2B) what mask_cond(force_mask=True) leaves
        if (c->cfg.cond_mode == RGN_COND_ACTION) {
            RGN_LAUNCH(c, KC_EMBED, s, launch_cond_rows(c->dp<float>(c->off_action), action, c->condemb, B, d, c->cfg.num_actions, s));
            RGN_LAUNCH(c, KC_EMBED, s, launch_fill_rows(c->condemb + (size_t)B * d, nullptr, B, d, s));
        } else if (c->cfg.cond_mode == RGN_COND_TEXT) {
            GemmArgs t = gemm_args(c, c->lin_text, text_feat, c->cfg.clip_dim, c->condemb, d, B);
            RGN_LAUNCH(c, KC_GEMM, s, launch_gemm(t, small_prec(c), s));
            RGN_LAUNCH(c, KC_EMBED, s, launch_fill_rows(c->condemb + (size_t)B * d, c->dp<float>(c->off_bt), B, d, s));  // embed_text(0) = bias
        }
        if (c->cfg.cond_mode != RGN_COND_NONE) {   // folded cross-attention image of the condition rows (cond | uncond)
            GemmArgs cg = gemm_args(c, c->lin_g, c->condemb, d, c->call_cond, c->L * d, 2 * B);
            cg.bias = nullptr;
            RGN_LAUNCH(c, KC_GEMM, s, launch_gemm(cg, small_prec(c), s));
        }
        c->cond_has_scale = scale != nullptr;
        if (scale) RGN_HIP(c, hipMemcpyAsync(c->scale, scale, (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, s));
        c->B = B;
        c->have_cond = true;
        return stream_exit(c, us);
    });
}

int rgn_denoise(rgn_handle h, const float* x, const int64_t* t, int32_t flags, float* out, void* stream) {
    return rgn_guard(h, "rgn_denoise", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        rgn_ctx* c = h;
        if (!c->have_cond) return c->fail(RGN_ERR_STATE, "rgn_denoise: no condition bound (rgn_set_condition)");
        if (!x || !t || !out) return c->fail(RGN_ERR_INVALID_ARG, "rgn_denoise: null pointer");
        const bool guided = flags & RGN_FLAG_GUIDED, uncond = flags & RGN_FLAG_UNCOND;
        if (guided && c->cfg.cond_mode == RGN_COND_NONE)
            return c->fail(RGN_ERR_INVALID_ARG, "rgn_denoise: guidance needs cond_mode text/action (cfg_sampler.py:26)");
        if (guided && !c->cond_has_scale) return c->fail(RGN_ERR_STATE, "rgn_denoise: guided evaluation needs y['scale']");
        hipStream_t us = reinterpret_cast<hipStream_t>(stream), s = c->stream;
        RGN_HIP(c, hipSetDevice(c->cfg.device));
        int rc = stream_enter(c, us);
        if (rc) return rc;
        const Dims dm = make_dims(c, c->B, guided);
        SampleParams sp{};
        sp.x0_out = out;
        sp.t_ext = t;
        sp.mode = 1;
        sp.guided = guided;
        RGN_HIP(c, hipMemcpyAsync(c->d_sp, &sp, sizeof(sp), hipMemcpyHostToDevice, s));
        rc = pack_state(c, x, dm, guided, s);
        if (rc) return rc;
        c->phase_x3 = true;               // a single evaluation is always split-bf16 under the precision schedule
        rc = run_eval(c, c->B, guided, uncond, false, s);
        if (rc) return rc;
        return stream_exit(c, us);
    });
}

int rgn_sample_range(rgn_handle h, int32_t sampler, int32_t guided, float eta, float* x, const float* noise, uint64_t seed,
                     uint64_t sample_offset, int32_t first_index, int32_t count, float* x0_out, int32_t use_graph,
                     int32_t clip_denoised, void* stream) {
    return rgn_guard(h, "rgn_sample_range", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        rgn_ctx* c = h;
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
    });
}

int rgn_set_x3_tail(rgn_handle h, int32_t tail_steps) {
    return rgn_guard(h, "rgn_set_x3_tail", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        if (tail_steps < -1) return h->fail(RGN_ERR_INVALID_ARG, "rgn_set_x3_tail: tail_steps < -1");
        h->x3_tail = tail_steps;
        return RGN_OK;
    });
}

int rgn_set_f16_steps(rgn_handle h, int32_t steps) {
    return rgn_guard(h, "rgn_set_f16_steps", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        if (steps < -1) return h->fail(RGN_ERR_INVALID_ARG, "rgn_set_f16_steps: steps < -1");
        h->f16_steps = steps;
        return RGN_OK;
    });
}

int rgn_precision_plan(rgn_handle h, int32_t B, int32_t guided, int32_t* f16_steps, int32_t* x3_tail) {
    return rgn_guard(h, "rgn_precision_plan", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        if (!f16_steps || !x3_tail) return h->fail(RGN_ERR_INVALID_ARG, "rgn_precision_plan: null output");
        if (!h->finalized) return h->fail(RGN_ERR_STATE, "rgn_precision_plan: weights not finalized");
        if (!h->have_sched) return h->fail(RGN_ERR_STATE, "rgn_precision_plan: no schedule (rgn_set_schedule)");
        if (B <= 0 || B > h->cfg.max_batch) return h->fail(RGN_ERR_INVALID_ARG, "rgn_precision_plan: B outside (0, max_batch]");
        const PrecPlan pp = prec_plan(h, make_dims(h, B, guided != 0), guided != 0);
        *f16_steps = pp.n16;
        *x3_tail = h->cfg.precision == RGN_PREC_BF16X3 ? h->S : pp.tail;
        return RGN_OK;
    });
}

int rgn_set_const_noise(rgn_handle h, int32_t on) {
    return rgn_guard(h, "rgn_set_const_noise", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        h->const_noise = on != 0;
        return RGN_OK;
    });
}

int rgn_set_small_batch_rows(rgn_handle h, int32_t rows) {
    return rgn_guard(h, "rgn_set_small_batch_rows", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        if (rows < -1) return h->fail(RGN_ERR_INVALID_ARG, "rgn_set_small_batch_rows: rows < -1");
        const int v = rows < 0 ? h->sb_rows_default : rows;
        if (v != h->sb_rows) {   // captured graphs hold the kernels of the engine that was selected when they were recorded
            if (h->stream) RGN_HIP(h, hipStreamSynchronize(h->stream));
            for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
            h->graphs.clear();
            h->sb_rows = v;
        }
        return RGN_OK;
    });
}

int rgn_set_option(rgn_handle h, const char* key, int32_t value) {
    return rgn_guard(h, "rgn_set_option", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        if (!key || !*key) return h->fail(RGN_ERR_INVALID_ARG, "rgn_set_option: empty key");
        if (h->finalized) return h->fail(RGN_ERR_STATE, "rgn_set_option: the switches select kernels when the weights are packed - set them before rgn_finalize_weights");
        static const char* known[] = {"NO_FUSED_QKV", "BIG_TILE_ROWS", "NO_ROWGEMM", "NO_MLP", "MLP_X3", "NO_QKV_RS", "NO_STEP_FUSION", "LAYERS_MIN_TQ", "LAYERS", "LAYERS_STEPS",
                                      "LAYERS_MIN_B", "LAYERS_GUIDED", "STEP_NO_QUADS", "NO_QKV_LONG", "SB_FUSED_ATTN", "SB_ROWS", "BULK_RESID_LO", "GRAPH_STEPS", "STREAMS", "SB_GRAPH",
                                      "BULK_F16", "F16_STEPS"};
        bool ok = false;
        for (const char* k : known) ok = ok || strcmp(k, key) == 0;
        if (!ok) return h->fail(RGN_ERR_BAD_KEY, std::string("rgn_set_option: unknown switch '") + key + "'");
        h->opts[key] = value;
        return RGN_OK;
    });
}

int rgn_set_layers_min_b(rgn_handle h, int32_t samples) {
    return rgn_guard(h, "rgn_set_layers_min_b", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        if (samples < -1) return h->fail(RGN_ERR_INVALID_ARG, "rgn_set_layers_min_b: samples < -1");
        const int v = samples < 0 ? h->layers_min_b_default : (samples < 1 ? 1 : samples);
        if (v != h->layers_min_b) {   // captured graphs hold the kernels of the form that was selected when they were recorded
            if (h->stream) RGN_HIP(h, hipStreamSynchronize(h->stream));
            for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
            h->graphs.clear();
            h->layers_min_b = v;
        }
        return RGN_OK;
    });
}

int rgn_plan_query(rgn_handle h, int32_t B, int32_t guided, int32_t split_phase, int32_t idx, const char** name, const char** kernel,
                   double* launches_per_eval, double* algo_flops_per_eval, double* l2_bytes_per_eval) {
    return rgn_guard(h, "rgn_plan_query", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        rgn_ctx* c = h;
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
                    case AF_QKV: mac[KC_QKV] = qkv + attn; n[KC_QKV] = L; kn[KC_QKV] = (x3 || !c->qkv_rs) ? "k_qkv_attn" : "k_qkv_attn_rs"; break;
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
    });
}

int rgn_randn_step(rgn_handle h, float* x, int32_t B, uint64_t seed, uint64_t sample_offset, int32_t loop_index, void* stream) {
    return rgn_guard(h, "rgn_randn_step", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        if (!x || B <= 0) return h->fail(RGN_ERR_INVALID_ARG, "rgn_randn: null x or B <= 0");
        if (loop_index < -1) return h->fail(RGN_ERR_INVALID_ARG, "rgn_randn_step: loop_index < -1");
        RGN_HIP(h, hipSetDevice(h->cfg.device));
        if (!h->finalized) return h->fail(RGN_ERR_STATE, "rgn_randn: weights not finalized");
        hipStream_t us = reinterpret_cast<hipStream_t>(stream), s = h->stream;
        int rc = stream_enter(h, us);
        if (rc) return rc;
        // Philox stream word: the loop index of the step the noise belongs to; 0xFFFFFFFF (loop_index -1) is the x_T draw
        RGN_LAUNCH(h, KC_UPDATE, s, launch_randn(x, B, h->F * h->cfg.num_frames, h->cfg.num_frames, seed, sample_offset, (uint32_t)loop_index, s));
        return stream_exit(h, us);
    });
}

int rgn_randn(rgn_handle h, float* x, int32_t B, uint64_t seed, uint64_t sample_offset, void* stream) {
    return rgn_guard(h, "rgn_randn", [&]() -> int {
        return rgn_randn_step(h, x, B, seed, sample_offset, -1, stream);
    });
}

int rgn_rot6d_to_matrix(rgn_handle h, const float* d6, float* mat, int64_t n, void* stream) {
    return rgn_guard(h, "rgn_rot6d_to_matrix", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        if (n < 0 || (n > 0 && (!d6 || !mat))) return h->fail(RGN_ERR_INVALID_ARG, "rgn_rot6d_to_matrix: bad argument");
        RGN_HIP(h, hipSetDevice(h->cfg.device));
        if (!h->finalized) return h->fail(RGN_ERR_STATE, "rgn_rot6d_to_matrix: weights not finalized");
        hipStream_t us = reinterpret_cast<hipStream_t>(stream), s = h->stream;
        int rc = stream_enter(h, us);
        if (rc) return rc;
        RGN_LAUNCH(h, KC_MISC, s, launch_rot6d(d6, mat, n, s));
        return stream_exit(h, us);
    });
}

int rgn_gaussian_filter1d(rgn_handle h, const float* x, float* out, int64_t rows, int32_t T, float sigma, void* stream) {
    return rgn_guard(h, "rgn_gaussian_filter1d", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        if (rows < 0 || T <= 0 || !(sigma > 0.f) || (rows > 0 && (!x || !out)))
            return h->fail(RGN_ERR_INVALID_ARG, "rgn_gaussian_filter1d: bad argument");
        RGN_HIP(h, hipSetDevice(h->cfg.device));
        if (!h->finalized) return h->fail(RGN_ERR_STATE, "rgn_gaussian_filter1d: weights not finalized");
        hipStream_t us = reinterpret_cast<hipStream_t>(stream), s = h->stream;
        int rc = stream_enter(h, us);
        if (rc) return rc;
        RGN_LAUNCH(h, KC_MISC, s, launch_gauss1d(x, out, rows, T, sigma, s));
        return stream_exit(h, us);
    });
}

int rgn_profile_enable(rgn_handle h, int32_t on) {
    return rgn_guard(h, "rgn_profile_enable", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        (void)hipSetDevice(h->cfg.device);
        (void)hipDeviceSynchronize();
        if (on && h->prof_pool.empty()) {
            h->prof_pool.resize(1024);
            for (auto& e : h->prof_pool) {
                RGN_HIP(h, hipEventCreate(&e.a));
                RGN_HIP(h, hipEventCreate(&e.b));
            }
        }
        h->prof_used = 0;
        for (int i = 0; i < KC_COUNT; ++i) {
            h->prof_ms[i] = 0;
            h->prof_n[i] = 0;
        }
        if (on && h->prof_bracket_ms < 0) {
            // What an event pair adds around ANY kernel (dispatch + event latency): the same bracket around a one-thread
            // no-op kernel, median of 64. rgn_profile_query reports it so that callers can subtract it per launch.
            std::vector<float> v;
            for (int i = 0; i < 64 && i < (int)h->prof_pool.size(); ++i) {
                (void)hipEventRecord(h->prof_pool[i].a, h->stream);
                (void)launch_advance(h->d_step + 3, h->stream);      // scratch slot of d_step[4]
                (void)hipEventRecord(h->prof_pool[i].b, h->stream);
            }
            (void)hipStreamSynchronize(h->stream);
            for (int i = 0; i < 64 && i < (int)h->prof_pool.size(); ++i) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, h->prof_pool[i].a, h->prof_pool[i].b) == hipSuccess) v.push_back(ms);
            }
            std::sort(v.begin(), v.end());
            h->prof_bracket_ms = v.empty() ? 0.0 : v[v.size() / 2];
        }
        h->prof = on != 0;
        return RGN_OK;
    });
}

int rgn_profile_bracket_overhead(rgn_handle h, double* ms) {
    return rgn_guard(h, "rgn_profile_bracket_overhead", [&]() -> int {
        if (!h || !ms) return RGN_ERR_INVALID_ARG;
        *ms = h->prof_bracket_ms < 0 ? 0.0 : h->prof_bracket_ms;
        return RGN_OK;
    });
}

int rgn_profile_query(rgn_handle h, int32_t idx, const char** name, double* total_ms, int64_t* launches) {
    return rgn_guard(h, "rgn_profile_query", [&]() -> int {
        if (!h) return RGN_ERR_INVALID_ARG;
        if (idx < 0 || idx >= KC_COUNT || !name || !total_ms || !launches) return h->fail(RGN_ERR_INVALID_ARG, "rgn_profile_query: bad argument");
        if (h->prof_used > 0) {
            (void)hipSetDevice(h->cfg.device);
            (void)hipDeviceSynchronize();
            for (size_t i = 0; i < h->prof_used; ++i) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, h->prof_pool[i].a, h->prof_pool[i].b) == hipSuccess) {
                    h->prof_ms[h->prof_pool[i].kc] += ms;
                    h->prof_n[h->prof_pool[i].kc] += 1;
                }
            }
            h->prof_used = 0;
        }
        *name = kclass_names[idx];
        *total_ms = h->prof_ms[idx];
        *launches = h->prof_n[idx];
        return RGN_OK;
    });
}

}  // extern "C"
