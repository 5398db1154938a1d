// Host side of libregennet_hip.so, shared declarations: the engine context behind rgn_handle and what the three host translation units exchange -
//   rgn_pack.cpp   checkpoint ingestion: reference state_dict -> folded / packed weight blob + workspace (rgn_finalize_weights)
//   rgn_plan.cpp   WHICH kernels run an evaluation / a sampling call and in which arithmetic (plan_eval, prec_plan), and their dispatch
//                  (run_eval, sample_range, plan_query)
//   rgn_abi.cpp    the C-ABI of include/regennet_hip.h: argument checks, the exception guard, the small entry points
// Not part of the C-ABI. All tensor arithmetic happens in the .hip files (rgn_internal.h declares their launchers).
#pragma once
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

namespace rgnh {

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

}  // namespace rgnh

struct rgn_ctx {
    rgn_config cfg{};
    std::string err;
    std::map<std::string, std::vector<int64_t>> expected;  // key -> shape (pe: shape[0] free)
    std::map<std::string, rgnh::HostTensor> sd;
    std::map<std::string, int> opts;   // rgn_set_option: per-handle kernel-selection switches (they take precedence over REGENNET_<KEY> in the environment)
    bool finalized = false, have_sched = false, have_cond = false;
    int F = 0, d = 0, Tq = 0, etd = 0, L = 0, H = 0, ff = 0, pe_len = 0;

    // packed weights
    std::vector<char> hblob;
    char* dblob = nullptr;
    size_t blob_bytes = 0;
    rgnh::Lin lin_x, lin_c, lin_t0, lin_t2, lin_g, lin_out, lin_text;
    std::vector<rgnh::LayerW> layers;
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
    // ... and guided runs: a MOTION per workgroup (its two evaluations back to back, x0_c parked meanwhile) or an EVALUATION per workgroup per step (k_layers<false>
    // over the 2 B rows + the guided k_step). The first fills the chip only when B alone does: at 2 B <= #CUs the second runs a step in ONE evaluation's latency
    // instead of two (cfg3's shape, same box: B = 64 1340 vs 786 motions/s, B = 128 2136 vs 1551; B = 160 1628 vs 1909, B = 256 2226 vs 2392).
    // LAYERS_GUIDED: 0 never a motion per workgroup | 1 (default) when 2 B > #CUs | 2 always; rgn_set_option accepts it after finalize too (-1: the default again)
    int layers_guided = 1, layers_guided_default = 1;
    int num_cus = 256;
    bool skip_embed_out = false;       // (set by run_eval around run_layers while it enqueues a fused step)
    int step_no_quads = 0;             // REGENNET_STEP_NO_QUADS=1 (tests)
    bool qkv_rs = true;                // plain-bf16 phase: k_qkv_attn with register-streamed weights (REGENNET_NO_QKV_RS=1: the DMA-fed loop)
    bool qkv_x3_dma = false;           // split-bf16 phase: keep the direct-to-LDS k_qkv_attn<true> (REGENNET_QKV_X3_DMA=1) instead of k_qkv_attn_rs_x3
    bool qkv_long = false;             // plain-bf16 phase, 65 .. 160 tokens: fused in_proj + attention per (sample, head) (REGENNET_NO_QKV_LONG=1: in_proj GEMM + k_attn_x3)
    bool sb = false;                   // small-batch engine: column-split GEMMs with consumer-side LayerNorm (k_sb_gemm)
    bool sb_attn = false;              // small-batch engine: in_proj + attention as one launch per layer (k_sb_qkv_attn; REGENNET_SB_FUSED_ATTN=1: it loses below B ~ 6)
    int sb_rows = 640;                 // evaluations of at most this many token rows take it (rgn_set_small_batch_rows; 0 disables)
    int sb_rows_default = 640;         // (REGENNET_SB_ROWS): measured crossover with the throughput kernels at 60 tokens: between B = 10 and 11
                                       // (round 3, 250-step calls: B = 10 108 vs 116 ms, B = 11 122 vs 117, B = 12 123 vs 117; it was B = 12 .. 16 in round 2)
    rgn::StepCoef* d_tab = nullptr;
    int* d_step = nullptr;
    rgn::SampleParams* d_sp = nullptr;
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
    int f16_steps_default = -1;        // (REGENNET_F16_STEPS / rgn_set_option "F16_STEPS")
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
    std::vector<rgnh::ProfEv> prof_pool;     // pre-created event pairs
    size_t prof_used = 0;
    double prof_ms[rgn::KC_COUNT] = {0};
    double prof_bracket_ms = -1.0;     // event-pair time around a no-op kernel (calibrated on first enable)
    int64_t prof_n[rgn::KC_COUNT] = {0};

    int fail(int code, const std::string& m) {
        err = m;
        return code;
    }
    template <typename T>
    T* dp(size_t off) const { return reinterpret_cast<T*>(dblob + off); }
};

namespace rgnh {
using namespace rgn;

extern const char* const kclass_names[KC_COUNT];

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

// ---- rgn_abi.cpp
int stream_enter(rgn_ctx* c, hipStream_t user);
int stream_exit(rgn_ctx* c, hipStream_t user);
// ---- rgn_pack.cpp
void build_expected(rgn_ctx* c);
int finalize_weights(rgn_ctx* c);                       // the body of rgn_finalize_weights
bool opt_get(const rgn_ctx* c, const char* key, int* value);
bool opt_flag(const rgn_ctx* c, const char* key);
// ---- rgn_plan.cpp
Dims make_dims(const rgn_ctx* c, int B, bool guided);
GemmArgs gemm_args(const rgn_ctx* c, const Lin& L, const float* A, int lda, float* C, int ldc, int M);
int pack_state(rgn_ctx* c, const float* x, const Dims& dm, bool guided, hipStream_t s);
int run_eval(rgn_ctx* c, int B, bool guided, bool uncond, bool sampling, hipStream_t s);
int build_step_table(rgn_ctx* c, float eta);
int sample_range(rgn_ctx* c, int32_t sampler, int32_t guided, float eta, float* x, const float* noise, uint64_t seed, uint64_t sample_offset,
                 int32_t first_index, int32_t count, float* x0_out, int32_t use_graph, int32_t clip_denoised, void* stream);   // the body of rgn_sample_range
int plan_query(rgn_ctx* c, int32_t B, int32_t guided, int32_t split_phase, int32_t idx, const char** name, const char** kernel,
               double* launches_per_eval, double* algo_flops_per_eval, double* l2_bytes_per_eval);                         // ... of rgn_plan_query
struct PrecPlan { int tail = 0, n16 = 0; };
PrecPlan prec_plan(const rgn_ctx* c, const Dims& dm, bool guided);

// precision of the per-schedule / per-condition / rgn_denoise-only GEMMs (k_gemm_f32 / k_gemm_bf16): the schedule mode
// runs them split-bf16 (they are once-per-call work)
inline int small_prec(const rgn_ctx* c) { return c->cfg.precision == RGN_PREC_BF16_X3TAIL ? RGN_PREC_BF16X3 : c->cfg.precision; }
// split-bf16 (three MFMAs per product) for the evaluation being enqueued?
inline bool eval_x3(const rgn_ctx* c) {
    return c->cfg.precision == RGN_PREC_BF16X3 || (c->cfg.precision == RGN_PREC_BF16_X3TAIL && c->phase_x3);
}
// do activation planes carry a lo part at all (allocation, sampler state, residual stream)?
inline bool has_lo(const rgn_ctx* c) { return c->cfg.precision == RGN_PREC_BF16X3 || c->cfg.precision == RGN_PREC_BF16_X3TAIL; }
inline bool use_sb(const rgn_ctx* c, int rows) { return c->sb && c->cfg.precision != RGN_PREC_F32 && rows <= c->sb_rows; }

}  // namespace rgnh
