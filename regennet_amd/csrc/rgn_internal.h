// Internal declarations shared by the host translation units (rgn_pack.cpp, rgn_plan.cpp, rgn_abi.cpp: rgn_host.h) and rgn_kernels.hip (device code).
// Not part of the C-ABI (that is include/regennet_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rgn {

// ---- kernel classes (for per-class HIP-event timing, rgn_profile_query) -------------------------
enum KClass : int {
    KC_GEMM = 0,      // MFMA GEMM launches (k_gemm_x3 / k_gemm_f32 / k_gemm_bf16; dominant)
    KC_ATTN,          // causal self-attention
    KC_LN,            // residual LayerNorm kernels
    KC_EMBED,         // pe gather / emb rows
    KC_UPDATE,        // pack / sampler update / output transpose / philox
    KC_MISC,
    KC_QKV,           // fused in_proj GEMM + attention (k_qkv_attn)
    KC_ROWLN,         // row-complete GEMM + residual + LayerNorm(s) (k_rowgemm<0>)
    KC_ROWACT,        // row-complete GEMM + activation (k_rowgemm<1>)
    KC_MLP,           // row-persistent layer tail (k_mlp)
    KC_SB,            // small-batch column-split GEMMs (k_sb_gemm)
    KC_STEP,          // fused step boundary: output projection + sampler update + next input embedding (k_step)
    KC_LAYERS,        // the whole decoder stack of an evaluation, one sample per workgroup (k_layers)
    KC_STEPS,         // a run of complete sampler steps (stack + step boundary per sample) in one launch (k_layers<true>)
    KC_COUNT
};

// ---- per-step scalar table (device resident; index = loop index i, S-1 .. 0) --------------------
// All coefficients are the reference's fp64 table entries cast to fp32 exactly like
// _extract_into_tensor does (gaussian_diffusion.py:1614), with the remaining scalar arithmetic of
// p_sample / ddim_sample done in fp32 on the host in the reference's operation order.
struct StepCoef {
    float c1, c2;        // DDPM: mean = c1*x0 + c2*x      (posterior_mean_coef1/2)
    float sig_ddpm;      // (t!=0) * exp(0.5*log_variance)
    float sr, srm1;      // DDIM: eps = (sr*x - x0)/srm1   (sqrt_recip / sqrt_recipm1 alphas_cumprod)
    float ca, cb;        // DDIM: mean = x0*ca + cb*eps    (sqrt(abar_prev), sqrt(1-abar_prev-sigma^2))
    float sig_ddim;      // (t!=0) * sigma(eta)
    int32_t t_model;     // timestep_map[i] : ORIGINAL index handed to the denoiser
    int32_t pad[3];
};

// ---- arguments that change between rgn_sample_range calls (device resident so a captured graph
//      of one step can be replayed with different bindings) ---------------------------------------
struct SampleParams {
    float* x;                 // [B,F,T] sampler state, updated in place
    const float* noise;       // [count,B,F,T] tape or nullptr (-> Philox)
    float* x0_out;            // optional pred_xstart [B,F,T]
    const int64_t* t_ext;     // rgn_denoise: external timesteps [B]; nullptr inside sampling loops
    unsigned long long seed;
    unsigned long long sample_offset;
    int32_t first_index;      // loop index that tape entry 0 belongs to
    int32_t sampler;          // RGN_SAMPLER_*
    int32_t mode;             // 0: sampler update, 1: output only (rgn_denoise)
    int32_t guided;
    int32_t clip;             // clamp pred_xstart to [-1,1]
    int32_t const_noise;      // p_sample's const_noise: every motion takes motion 0's per-step draw
};

struct GemmArgs {
    const float* A; int lda;          // [M,K] row-major activations (fp32)
    const float* W;                   // [N,Kp] row-major fp32 weights, K zero-padded to Kp (multiple of 16)
    const uint16_t* Whi;              // [N,Kp] bf16 high parts   (BF16X3 / BF16)
    const uint16_t* Wlo;              // [N,Kp] bf16 low parts    (BF16X3)
    const float* bias;                // [N] or nullptr
    const float* add; int ldadd;      // optional addend, row r uses add[(r % add_mod) * ldadd + n]
    int add_mod;                      // 0 -> r itself
    float* C; int ldc;
    int M, N, K, Kp;
    int act;                          // 0 none, 1 gelu(erf), 2 silu, 3 relu
};

// bf16 operand planes use the "K32-blocked" layout [Kp/32][rows][32] (see rgn_gemm_x3.hip).
struct GemmX3Args {
    const __bf16* Ahi; const __bf16* Alo; int a_rows;   // activation planes [Kp/32][a_rows][32], a_rows >= M
    const __bf16* Whi; const __bf16* Wlo;               // weight planes [Kp/32][N][32]
    const float* bias;
    const float* add; int ldadd;                        // optional fp32 addend [M, ldadd]
    float* C; int ldc;                                  // optional fp32 output [M, ldc]
    __bf16* Chi; __bf16* Clo; int c_rows;               // optional split output planes [ceil(N/32)][c_rows][32]
    const __bf16* Rhi; const __bf16* Rlo; int r_rows;   // launch_sg_tconv tail 2 only: residual planes [N/32][r_rows][32], added as (hi + lo)
    int poly_T, poly_V, poly_region;                    // launch_sg_tconv tail 3 only: frames per sequence of the rows (+ 4 pads), vertices, rows per output region
    // launch_sg_tconv_s2 only: a second product accumulated into the same tile, A2 [k2][a2_rows][32] (rows as the output's) x W2 [k2][N][32] - the
    // block's convolved shortcut
    const __bf16* A2hi; const __bf16* A2lo; int a2_rows; const __bf16* W2hi; const __bf16* W2lo; int k2;
    int M, N, Kp;
    int act;
    // Optional "attention-ready" output of the packed in_proj GEMM (N = 3*d): instead of C / Chi the epilogue
    // scatters q (pre-scaled by 1/sqrt(dh)), k and v as [Bm*H][Tqp][dh] split planes (one contiguous slab per head;
    // the Vt* pointers hold v, which k_attn_x3 transposes on its way into LDS).
    __bf16 *Qhi, *Qlo, *Khi, *Klo, *Vthi, *Vtlo;
    int d, H, dh, Tq, Tqp;
    unsigned tq_magic;                               // floor(2^32 / Tq) + 1: m / Tq == umulhi(m, tq_magic) for m * Tq < 2^32
    float qscale;
    // launch_gemm_x3_sg only (rgn_stgcn.hip): a temporal convolution as ONE GEMM over K = channel blocks x taps - k-block kt reads plane block
    // kt / a_taps at the byte offset a_tap[kt % a_taps] of its tap (a row shift for a stride-1 convolution; for stride 2 the even- / odd-frame
    // region of a polyphase plane plus a row shift; a_taps = 0: plain addressing; the planes carry zero pad frames and guard rows for the taps'
    // reach). TAPS INNERMOST: consecutive k-steps re-read the same channel block 56 rows further on, so the shifted windows hit L2 - with the
    // taps outermost every tap swept all channel blocks (256 KB per tile, 8 MB per XCD) before the next one touched the same lines again, and
    // the 256-channel blocks fetched 32 GB per forward over the fabric (L2 hit rate 0.61). The addend row is (row % add_mod) when add_mod > 0; act 3 = ReLU
    int a_taps, add_mod;
    long long a_tap[9];
    int f16;                                            // launch_sg_gcn / launch_sg_tconv(_s2) only: the single-plane fp16 form (rgn_sg_kernels.hip) - Ahi, Whi, Rhi,
                                                        // A2hi, W2hi and Chi hold IEEE fp16, the lo pointers are not used
};

// Split-bf16 activation planes in the K32-blocked layout; hi == nullptr means "not requested".
struct Planes {
    __bf16* hi;
    __bf16* lo;
    int rows;     // row count of the plane (the M of the consuming GEMM)
};

// Fused in_proj GEMM + causal self-attention, one workgroup per (sample pair, half of the heads) (rgn_qkv_attn.hip)
struct QkvAttnArgs {
    const __bf16* Ahi; const __bf16* Alo; int a_rows;   // layer input planes [Kp/32][a_rows][32] (advanced to the first sample)
    const __bf16* Whi; const __bf16* Wlo;               // in_proj weight planes [Kp/32][3d][32]
    const __bf16* Wfr;                                  // the hi plane in MFMA-fragment order [Kp/32][3d/32][2][64][8] (plain-bf16 phase, d = 512), nullable
    const __bf16* Wfr_lo;                               // split phase, register-streamed form (k_qkv_attn_rs_x3): the lo plane in the same order, with Wfr = the bf16 hi plane; nullable
    const float* bias;                                  // in_proj bias [3d]
    Planes out;                                         // attention output planes (advanced to the first sample's row)
    int Bm, Kp, d, H, Tq;
    float qscale;
    int Bm_eval;                                        // samples of the WHOLE evaluation (all kernel chains; 0: = Bm): what else runs beside this launch
    int f16;                                            // k_qkv_attn_long / k_qkv_attn_rs<1>: Ahi, Wfr and the output plane hold IEEE fp16 (OpFmt<true>), not bf16
};
bool qkv_attn_supported(int Tq, int dh, int d);
hipError_t configure_qkv_attn();
hipError_t launch_qkv_attn(const QkvAttnArgs& g, bool x3, hipStream_t s);
// long sequences (65 .. 160 tokens), plain-bf16 phase, d = 512: one workgroup per (sample, head) (rgn_qkv_attn_long.hip)
bool qkv_attn_long_supported(int Tq, int dh, int d);
hipError_t configure_qkv_attn_long();
hipError_t launch_qkv_attn_long(const QkvAttnArgs& g, hipStream_t s);

// Row-complete plain-bf16 GEMM with the layer tail fused (rgn_rowgemm.hip): 64 complete rows x 512 columns per workgroup,
// activation tile resident in LDS, weights streamed into registers.
struct RowGemmArgs {
    const __bf16* A; int a_rows;          // activation plane (hi) [Kp/32][a_rows][32], advanced to the first row
    const __bf16* W;                      // weight plane (hi), fragment-ordered [Kp/32][N/32][2][64][8] (see rgn_rowgemm.hip)
    const float* bias;                    // [N]
    int M, N, Kp;
    // ---- epilogue "act": out = act(A.W^T + bias) as bf16 planes (N % 32 == 0)
    int act;                              // 0 none, 1 gelu(erf), 2 packed in_proj -> attention-ready q (pre-scaled) / k / v planes
    __bf16 *Qhi, *Khi, *Vhi;              // act == 2: [Bm*H][Tqp][dh] planes (N = 3 * 512: one column chunk each)
    int H, dh, Tqp; float qscale;         // act == 2 (Tq below)
    const float* add; int ldadd;          // (reserved, must be null)
    float* C; int ldc;                    // (reserved, must be null)
    __bf16* Chi; __bf16* Clo; int c_rows; // output planes [N/32][c_rows][32]; Clo nullable
    // ---- epilogue "ln" (N == 512): out = LN_b(LN_a(A.W^T + bias + resid) + pervec[row / Tq] + stepvec[*d_step])
    const __bf16* Rhi; const __bf16* Rlo; int r_rows;   // residual planes (Rlo nullable); may alias the output planes
    __bf16* Ohi; __bf16* Olo; int o_rows;
    const float *ga, *ba, *gb, *bb;       // gb == nullptr: single norm
    const float* pervec; int ldper;
    const float* stepvec; int ldstep; const int* d_step;
    int Tq;
};
bool rowgemm_supported(int N, int Kp, bool ln);
hipError_t configure_rowgemm();
hipError_t launch_rowgemm(const RowGemmArgs& g, bool ln, hipStream_t s);

// Row-persistent decoder-layer tail (rgn_mlp2.hip): out_proj + norm1 + folded cross-attention + norm2 + linear1 + GELU + linear2 + norm3
// for 64-row tiles, plain-bf16 phase, d = 512, ff = 1024. All planes are hi-only K32-blocked [16][rows][32]; weights fragment-ordered.
struct MlpArgs {
    const __bf16* att;                    // attention output planes (A operand of out_proj), advanced to the first row
    const __bf16* h;                      // layer input planes (residual of norm1)
    __bf16* out;                          // layer output planes (may alias h)
    int rows;                             // plane row count (stride), M rows processed
    int M;
    const __bf16 *Wo, *W1, *W2;           // out_proj [512x512], linear1 [1024x512], linear2 [512x1024]
    const float *bo, *bf1, *bf2;
    const float *g1, *b1, *g2, *b2, *g3, *b3;
    const float* pervec; int ldper;       // + pervec[(row / Tq) * ldper + n]     (nullable)
    const float* stepvec; int ldstep; const int* d_step;   // + stepvec[(*d_step) * ldstep + n] (nullable)
    int Tq;
    int f16;                              // k_mlp2<2> only: att / h / out planes and the three weight planes hold IEEE fp16 (OpFmt<true>)
};
bool mlp_supported(int d, int ff, int Tq);
hipError_t configure_mlp();
hipError_t launch_mlp(const MlpArgs& g, hipStream_t s);
// the layer tail of the SPLIT-bf16 phase as one row-persistent kernel (rgn_mlp_x3.hip): 32-row tiles, (hi, lo) plane pairs everywhere, three MFMAs
// per product, two-pass LayerNorm, erf GELU - replaces k_gemm_x3 x 3 + k_layernorm x 2 per layer. Weight planes fragment-ordered, hi and lo.
struct MlpX3Args {
    MlpArgs p;                            // hi planes / fragment planes and the vectors, as for k_mlp2
    const __bf16 *att_lo, *h_lo;          // lo planes of the attention output and of the layer input (advanced like the hi planes)
    __bf16* out_lo;                       // lo plane of the layer output (may alias h_lo)
    const __bf16 *Wo_lo, *W1_lo, *W2_lo;  // lo fragment planes
};
bool mlp_x3_supported(int d, int ff, int Tq);
hipError_t configure_mlp_x3();
hipError_t launch_mlp_x3(const MlpX3Args& g, hipStream_t s);
// second build of the layer tail (rgn_mlp2.hip): rows = 64 (8 waves, one workgroup per CU) or 32 (4 waves, two per CU)
bool mlp2_supported(int rows, int d, int ff, int Tq);
hipError_t configure_mlp2();
hipError_t launch_mlp2(int rows, const MlpArgs& g, hipStream_t s);

// The whole decoder stack of one evaluation as one kernel, one sample (Tq <= 64 tokens) per workgroup (rgn_layers.hip): plain-bf16 phase,
// d = 512, ff = 1024, 4 heads of 128. Weight planes fragment-ordered as for k_mlp / k_qkv_attn_rs.
constexpr int LY_MAXL = 8;
struct StepCoef;
struct SampleParams;
struct LayerWts {                          // one entry per decoder layer; the table travels in the kernel arguments (scalar loads)
    const __bf16 *Wqkv, *Wo, *W1, *W2;
    const float *bqkv, *bo, *bf1, *bf2;
    const float *g1, *b1, *g2, *b2, *g3, *b3;
};
struct LayersArgs {
    const __bf16* h;                      // residual-stream planes (hi) [16][rows][32], advanced to the first sample's row
    __bf16* out;                          // (may alias h)
    int rows;                             // plane row count (stride)
    int Bm, Tq, L;                        // samples of this launch, tokens per sample, layers
    LayerWts lw[LY_MAXL];                 // [L], L <= LY_MAXL
    const float* pervec; int ldper;       // + pervec[sample * ldper + layer * 512 + n]   (nullable; advanced to the first sample)
    const float* stepvec; int ldstep; const int* d_step;   // + stepvec[(*d_step) * ldstep + layer * 512 + n] (nullable)
    float qscale;
    // ---- steps > 0: the kernel runs `steps` complete sampler steps per sample (unguided): after the stack the step boundary of
    //      rgn_step.hip in per-sample form - output projection, sampler update of x in place, the next evaluation's input embedding
    //      into the resident image - and the loop index *d_step moves on by `steps` when the last workgroup finishes
    int steps;
    const __bf16* Wout; const float* bout; int F, nb_out;    // output projection, fragment-ordered [16][nb_out][2][64][8]
    const __bf16* Wx;                                         // input embedding (folded), fragment-ordered [11][16][2][64][8]
    const __bf16* c0;                                         // hoisted condition part [rows, 512] as bf16, advanced like h
    const StepCoef* tab; int* d_stepw; const SampleParams* sp;
    int B, s0;                                                // motions in the bound condition, first sample of this launch
    int no_quads;
    // ---- scale != nullptr: classifier-free guidance inside the launch (cfg_sampler.py:22-31). A workgroup owns a MOTION: its conditional
    //      evaluation (planes rows of sample b) and its unconditional one (rows of sample B + b) run back to back in the same workgroup, the
    //      conditional x0 parked in `park` meanwhile; Bm = B motions; pervec / c0 / h hold both halves, `half` = row distance B * T
    const float* scale; int half; float* park;                // scale[B]; park: >= B * 6 * 4096 floats of scratch
    // ---- f16 != 0: every 16-bit operand is IEEE fp16 instead of bf16 - the weight planes (same fragment order), c0, and the
    //      residual-stream planes h / out (same K32-blocked layout): v_mfma_f32_32x32x16_f16 runs at the bf16 instruction's rate with 2^-12
    //      operand rounding instead of 2^-9 (rgn_set_option "BULK_F16")
    int f16;
};
bool layers_supported(int d, int ff, int H, int Tq, int L);
bool layers_steps_supported(int d, int F, int Kpx);
hipError_t configure_layers();
hipError_t launch_layers(const LayersArgs& g, hipStream_t s);

// Small-batch column-split GEMM (rgn_sb.hip): 64 rows x 32 output columns per workgroup; LayerNorm applied by the consumer.
struct SbArgs {
    // PRE 1: A = LN_b(LN_a(src) + stepvec[*d_step] + pervec[row / Tq]) of fp32 rows [M, 512]; ga == nullptr: A = src; gb == nullptr: one norm
    const float* src;
    const float *ga, *ba, *gb, *bb;
    const float* pervec; int ldper;
    const float* stepvec; int ldstep; const int* d_step;
    int Tq;                                             // tokens per sample (pervec row, q/k/v scatter)
    float* xout;                                        // normalised rows [M, 512] (written by the first 16 column slices), nullable
    // PRE 0: A = K32-blocked planes [Kp/32][a_rows][32]
    const __bf16 *Ahi, *Alo; int a_rows;
    const __bf16 *Whi, *Wlo; int w_rows;                // weight planes [Kp/32][w_rows][32]
    const float* bias;                                  // [N], nullable
    int M, N, Kp;
    // POST 0: C = A.W^T + bias + resid (fp32 rows)
    const float* resid; int ldr;
    float* C; int ldc;
    // POST 1: gelu(A.W^T + bias) as K32-blocked planes [N/32][c_rows][32]
    __bf16 *Chi, *Clo; int c_rows;
    // POST 2: packed in_proj -> attention-ready q (pre-scaled) / k / v planes [Bm*H][Tqp][dh]
    __bf16 *Qhi, *Qlo, *Khi, *Klo, *Vhi, *Vlo;
    int d, H, dh, Tqp; float qscale;
    // k_sb_qkv_attn (rgn_sb_attn.hip): the attention output, K32-blocked planes of the out_proj GEMM
    Planes att;
};
bool sb_supported(int d, int ff, int dh);
hipError_t configure_sb();
hipError_t launch_sb_gemm(const SbArgs& g, int pre, int post, bool x3, hipStream_t s);
// LayerNorm prologue + in_proj + causal self-attention of one (sample, head) per workgroup (rgn_sb_attn.hip): d = 512, dh = 128, <= 64 tokens
bool sb_qkv_attn_supported(int d, int dh, int Tq);
hipError_t configure_sb_qkv_attn();
hipError_t launch_sb_qkv_attn(const SbArgs& g, int Bm, bool x3, hipStream_t s);

// The 16-bit operand format of the plain phase's MFMAs (weights, activation images / planes, q / k / v / p): bf16 (8 mantissa bits) or IEEE fp16
// (11). Both instructions are 8 passes of 4 cycles per 32 x 32 x 16 tile and take 16 bytes per lane and operand, so a kernel's structure - rings,
// images, waits - does not depend on the format; what changes is every operand's rounding (2^-9 -> 2^-12 relative) and its range (fp16: 6.1e-5 ..
// 65504 normal; LayerNorm outputs, GELU values, softmax probabilities and the weights of a transformer sit well inside, and rgn_finalize_weights
// refuses a checkpoint that does not). Accumulation, LayerNorm statistics, softmax and the sampler update are fp32 either way. Nominally the same
// rate - but the chip is power-managed under a matrix load and a pure f16 MFMA loop sustains 7.5 - 8 % less than the bf16 one
// (tools/experiments/mfma_sustained.hip), which is why fp16 is a PHASE of the precision schedule (rgn_set_f16_steps), not its plain format.
template <bool F16> struct OpFmt;
template <> struct OpFmt<false> {
    typedef __bf16 t;
    typedef __bf16 v8 __attribute__((ext_vector_type(8)));
    typedef __bf16 v4 __attribute__((ext_vector_type(4)));
    typedef float acc16 __attribute__((ext_vector_type(16)));
    static __device__ __forceinline__ acc16 mfma(v8 a, v8 b, acc16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct OpFmt<true> {
    typedef _Float16 t;
    typedef _Float16 v8 __attribute__((ext_vector_type(8)));
    typedef _Float16 v4 __attribute__((ext_vector_type(4)));
    typedef float acc16 __attribute__((ext_vector_type(16)));
    static __device__ __forceinline__ acc16 mfma(v8 a, v8 b, acc16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// XCD-affine workgroup order: the hardware places workgroup id b on XCD b % 8. Remapping the id so that every XCD gets one
// CONTIGUOUS range of tiles / samples makes the rows a kernel reads the rows the previous kernel of the chain wrote on the
// same XCD (k_qkv_attn -> k_mlp -> k_qkv_attn ...): they are still in that XCD's L2 instead of behind the fabric.
__device__ __forceinline__ int xcd_affine(int bid, int nwg) {
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
}

// The other half-wave's value (lane ^ 32): v_permlane32_swap_b32 a, b gives a' = [a.lo | b.lo], b' = [a.hi | b.hi], so with b a copy of a in a
// register of its own b' and a' are the two halves' values in every lane - one VALU instruction where __shfl_xor(v, 32) is a ds_bpermute round trip.
// (inline asm: the builtin's second result is miscompiled by ROCm 7.2's clang, it adds a' to itself; tools/permlane_check.hip)
__device__ __forceinline__ void half_swap(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float half_max(float v) {
    float o = v;
    asm volatile("" : "+v"(o));
    half_swap(v, o);
    return fmaxf(v, o);
}
__device__ __forceinline__ float half_sum(float v) {
    float o = v;
    asm volatile("" : "+v"(o));
    half_swap(v, o);
    return v + o;
}

// Step boundary of the plain-bf16 phase as one kernel (rgn_step.hip): output projection + sampler update + the next
// evaluation's input embedding for 64-row tiles; d = 512, no emb_trans_dec token; with or without guidance.
struct StepCoef;
struct SampleParams;
struct StepArgs {
    const __bf16* h;            // last layer's output planes (hi) [16][rows][32], advanced to the first row of this launch
    __bf16* hout;               // residual-stream planes the next evaluation's layer 0 reads (normally the same buffer)
    int rows, M;                // plane row count; token rows of this launch
    const __bf16* Wout; const float* bout; int F, nb_out;    // output projection, fragment-ordered [16][nb_out][2][64][8] (rows zero-padded)
    const __bf16* Wx; int nkx;  // input embedding (folded), fragment-ordered [nkx][16][2][64][8]
    const __bf16* c0;           // hoisted condition part [M, 512] as bf16 (advanced): 16 instead of 31 MB per step at B = 256
    const StepCoef* tab; int* d_step; const SampleParams* sp;
    int T, B, s0;               // frames (= tokens) per sample, motions in the bound condition, first sample of this launch
    int total_tiles;            // 64-row tiles over ALL launches of the step (loop-index ticket)
    int no_quads;               // tests: draw the noise per element (philox_normal) instead of per quad of lanes (bit-identical)
    const float* scale; int half;   // guided sampling: scale[B] (nullptr: unguided) and the row distance B * T from a token's conditional
                                    // to its unconditional row (h / hout / c0 hold both halves; M counts the conditional rows)
    int f16;                        // h / hout planes, Wout, Wx and c0 hold IEEE fp16 (OpFmt<true>)
};
bool step_fused_supported(int d, int F, int Kpx);
hipError_t configure_step();
hipError_t launch_step(const StepArgs& g, hipStream_t s);

struct Dims {
    int B;        // motions in the bound condition
    int Bm;       // rows of the batched evaluation (B, or 2B under guidance)
    int T, Tq;    // frames, tokens per sample (T + emb_trans_dec)
    int etd;      // emb_trans_dec as 0/1 (row offset of frame 0 inside a sample)
    int F;        // njoints*nfeats
    int d, H, dh, ff, L;
};

// ---- launchers (rgn_kernels.hip) ----------------------------------------------------------------
hipError_t launch_gemm(const GemmArgs& g, int precision, hipStream_t s);
hipError_t launch_gemm_x3(const GemmX3Args& g, bool x3, int variant, hipStream_t s);
hipError_t configure_gemm_x3();
// split-bf16 GEMMs of the ST-GCN evaluator (rgn_stgcn.hip): 256-row tiles x the layer's channel count (64 / 128 / 256), shifted-row
// A addressing for the 9 x 1 temporal convolution, (row % V) addend, ReLU
hipError_t launch_gemm_x3_sg(const GemmX3Args& g, hipStream_t s);
hipError_t configure_gemm_x3_sg();
// ... and the stride-1 9 x 1 temporal convolution with the activation window resident in LDS (K order: channel block, tap): V = rows per frame
bool sg_tconv_supported(int N, int Kp, int V);
hipError_t launch_sg_tconv(const GemmX3Args& g, int V, int tail, bool small_tiles, hipStream_t s);   // tail 0: C = conv + bias (fp32) | 1: planes relu(conv + bias) | 2: planes relu(conv + bias + R) | 3: as 2, polyphase
hipError_t configure_sg_tconv();
// ... the stride-2 form on polyphase planes (region O starts o_rows rows behind region E; M = rows of one region = output rows), + the convolved shortcut;
// output: planes relu(conv + shortcut + bias)
bool sg_tconv_s2_supported(int N, int Kp, int V);
hipError_t launch_sg_tconv_s2(const GemmX3Args& g, int V, long long o_rows, hipStream_t s);
// ... and graph aggregation + 1 x 1 convolution as one kernel: A = the block's INPUT planes, Kp = KP x C_in. The nonzero lists of A'_k[:, w] come as 8
// slots per vertex, sl_v / sl_a [V][8] (source vertex, coefficient); slot s serves partition (slot_k >> 4 s) & 15 (15: unused) and a list shorter than
// its partition's slot count is padded with (w, 0)
bool sg_gcn_supported(int N, int Kp, int V, int KP);
hipError_t launch_sg_gcn(const GemmX3Args& g, int V, int KP, unsigned slot_k, const int* sl_v, const float* sl_a, int widest_tile, bool per_block_barrier, hipStream_t s);
hipError_t configure_sg_gcn();
hipError_t configure_attention(int Tq, int dh);
struct AttnX3Args {
    const __bf16 *Qhi, *Qlo, *Khi, *Klo, *Vthi, *Vtlo;   // layouts above
    Planes out;                                            // K32-blocked planes of the out_proj GEMM
    int Bm, H, dh, d, Tq, Tqp;
    bool x3;
};
bool attn_x3_supported(int Tq, int dh);
hipError_t configure_attn_x3(int Tq, int dh);
hipError_t launch_attn_x3(const AttnX3Args& a, hipStream_t s);
hipError_t launch_attention(const float* qkv, float* out, Planes op, const Dims& dm, hipStream_t s);
// h_out = LN_b( LN_a(in) + addvec[row/Tq] ) when ln_b != nullptr, else LN_a(in)
// addvec: per-sample vector (row/Tq)*ldadd, may be nullptr; stepvec: per-step vector at (*d_step)*ldstep, may be nullptr
hipError_t launch_layernorm(const float* in, Planes resid, float* out, Planes op, int M, int d, const float* ga, const float* ba,
                            const float* addvec, int ldadd, const float* stepvec, int ldstep, const int* d_step, int Tq,
                            const float* gb, const float* bb, hipStream_t s);
hipError_t launch_gather_pe(const float* pe, const StepCoef* tab, const int* d_step, const SampleParams* sp,
                            float* out, int Bm, int B, int d, int pe_len, hipStream_t s);
hipError_t launch_gather_pe_all(const float* pe, const StepCoef* tab, float* out, int S, int d, hipStream_t s);
hipError_t launch_emb_rows(const float* emb, const float* stepemb, const int* d_step, const float* pe, float* h, Planes hp,
                           const Dims& dm, int wo_pos, hipStream_t s);
hipError_t launch_add_pe(float* c0, const float* pe, const Dims& dm, hipStream_t s);
hipError_t launch_pack_x(const float* x, float* xin, Planes xp, int copies, const Dims& dm, hipStream_t s);
hipError_t launch_update(const float* x0tok, const float* scale, const StepCoef* tab, int* d_step,
                         const SampleParams* sp, float* xin, Planes xp, const Dims& dm, int b0, int nb, hipStream_t s);
hipError_t launch_advance(int* d_step, hipStream_t s);
hipError_t launch_cond_rows(const float* table, const int64_t* action, float* out, int B, int d, int num_actions, hipStream_t s);
hipError_t launch_fill_rows(float* out, const float* row, int rows, int d, hipStream_t s);
hipError_t launch_cvt_bf16(const float* in, __bf16* out, size_t n, hipStream_t s);
hipError_t launch_cvt_f16(const float* in, _Float16* out, size_t n, hipStream_t s);          // fp32 -> fp16 (rne)
hipError_t launch_bf16_to_f16(__bf16* inout, size_t n, hipStream_t s);                       // in place: a bf16 buffer re-encoded as fp16 (n % 8 == 0)
hipError_t launch_randn(float* x, int B, int FT, int T, unsigned long long seed, unsigned long long sample_offset,
                        uint32_t noise_stream, hipStream_t s);
hipError_t launch_rot6d(const float* d6, float* mat, long long n, hipStream_t s);
hipError_t launch_gauss1d(const float* x, float* out, long long rows, int T, float sigma, hipStream_t s);

}  // namespace rgn
