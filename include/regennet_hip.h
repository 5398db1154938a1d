/*
 * regennet_hip.h — C-ABI of libregennet_hip.so: the MI355X (gfx950) implementation of ReGenNet's
 * diffusion-sampling hot path. Plain C types only; device memory is passed as raw
 * pointers (e.g. torch.Tensor.data_ptr()), streams as hipStream_t cast to void*.
 *
 * The reference has no FFI of its own (it is pure Python/PyTorch); every entry point below states
 * the reference interface (file:line under the upstream repo) whose arithmetic it replaces. The
 * reference-side binding a maintainer would add is shown in INTEGRATION.md; the in-tree binding is
 * regennet_amd/_lib.py (ctypes).
 *
 * Conventions
 *   - every function returns 0 on success or a negative rgn_status; rgn_last_error() gives text.
 *   - no exceptions cross the boundary (every entry point catches at the boundary: RGN_ERR_INTERNAL);
 *     a handle is NOT thread-safe (one handle per device/stream).
 *   - "x" tensors are fp32 [B, njoints, nfeats, T] contiguous — the reference's boundary layout
 *     (model/cmdm.py:173-177); timesteps are int64 like the reference's `t` tensors.
 */
#ifndef REGENNET_HIP_H
#define REGENNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the entry points below are its ONLY dynamic symbols. */
#if defined(__GNUC__) || defined(__clang__)
#define RGN_API __attribute__((visibility("default")))
#else
#define RGN_API
#endif

typedef struct rgn_ctx* rgn_handle;

typedef enum {
    RGN_OK = 0,
    RGN_ERR_INVALID_ARG = -1,    /* bad pointer / size / enum                                   */
    RGN_ERR_BAD_KEY = -2,        /* unexpected state_dict key (utils/model_util.py:7)           */
    RGN_ERR_BAD_SHAPE = -3,      /* tensor shape does not match the configuration               */
    RGN_ERR_MISSING_KEY = -4,    /* finalize: a required key was never loaded (model_util.py:8) */
    RGN_ERR_STATE = -5,          /* call order violated (e.g. sample before set_schedule)       */
    RGN_ERR_HIP = -6,            /* HIP runtime error (text in rgn_last_error)                  */
    RGN_ERR_UNSUPPORTED = -7,    /* configuration outside the hot path (e.g. arch != online)    */
    RGN_ERR_INTERNAL = -8        /* a host-side C++ exception (std::bad_alloc ...) caught at the boundary */
} rgn_status;

enum { RGN_CM_ADD = 0, RGN_CM_CONCAT = 1 };                 /* --cm_mode, model/cmdm.py:207-211  */
enum { RGN_COND_NONE = 0, RGN_COND_ACTION = 1, RGN_COND_TEXT = 2 }; /* cond_mode, model_util.py:25-30 */
enum { RGN_PREC_F32 = 0,      /* fp32-input MFMA (v_mfma_f32_32x32x2_f32): exact fp32 products     */
       RGN_PREC_BF16X3 = 1,   /* split-bf16: a*b ~ ah*bh + ah*bl + al*bh on bf16 MFMA, fp32 accum  */
       RGN_PREC_BF16 = 2,     /* plain bf16 MFMA inputs (fastest; outside the 1e-3 parity bound)   */
       RGN_PREC_BF16_X3TAIL = 3 }; /* precision schedule (default): plain-bf16 GEMM operands while the sampler
                                 still contracts errors (large t), split-bf16 for the last steps of a sampling
                                 loop (rgn_set_x3_tail) and for every rgn_denoise call; the residual stream and
                                 LayerNorm/softmax stay hi+lo / fp32 throughout. Inside the 1e-3 parity bound. */
enum { RGN_SAMPLER_DDPM = 0,  /* GaussianDiffusion.p_sample   diffusion/gaussian_diffusion.py:508  */
       RGN_SAMPLER_DDIM = 1 };/* GaussianDiffusion.ddim_sample diffusion/gaussian_diffusion.py:744 */
enum { RGN_FLAG_UNCOND = 1,   /* y['uncond']=True  (model/cmdm.py:181, mask_cond :129-137)         */
       RGN_FLAG_GUIDED = 2 }; /* ClassifierFreeSampleModel.forward (model/cfg_sampler.py:24-31)    */

/* Mirrors the keyword arguments `CMDM(**get_model_args(args, data))` receives for arch='online'
 * (utils/model_util.py:20-72; model/cmdm.py:13-16). */
typedef struct {
    int32_t njoints;        /* 56 for SMPL-X (model_util.py:45-46)                     */
    int32_t nfeats;         /* 6 for rot6d (model_util.py:47-48)                       */
    int32_t num_frames;     /* T: 60 ntu / 150 chi3d (model_util.py:61-64)             */
    int32_t latent_dim;     /* d, --latent_dim (parser_util.py:126)                    */
    int32_t ff_size;        /* 1024 (model_util.py:69)                                 */
    int32_t num_heads;      /* 4 (model_util.py:69)                                    */
    int32_t num_layers;     /* --layers                                                */
    int32_t cm_mode;        /* RGN_CM_*                                                */
    int32_t cond_mode;      /* RGN_COND_*                                              */
    int32_t num_actions;    /* rows of embed_action.action_embedding (cmdm.py:361)     */
    int32_t clip_dim;       /* 512 (cmdm.py:15)                                        */
    int32_t emb_trans_dec;  /* --emb_trans_dec (cmdm.py:212-215,224-225)               */
    int32_t wo_pos_emb;     /* --wo_pos_emb (cmdm.py:217)                              */
    int32_t max_batch;      /* largest B any later call will use                       */
    int32_t precision;      /* RGN_PREC_*                                              */
    int32_t device;         /* HIP device ordinal                                      */
} rgn_config;

/* Per-timestep fp64 tables of the (re-spaced) diffusion, each of length S, exactly the attributes
 * GaussianDiffusion.__init__ builds (diffusion/gaussian_diffusion.py:172-209) after
 * SpacedDiffusion.__init__ re-derived the betas (diffusion/respace.py:73-87). */
typedef struct {
    int32_t S;                              /* num_timesteps after respacing                         */
    const int64_t* timestep_map;            /* respace.py:75,85 — ORIGINAL index of each kept step   */
    const double* posterior_mean_coef1;     /* gaussian_diffusion.py:199-201                          */
    const double* posterior_mean_coef2;     /* :202-206                                               */
    const double* model_log_variance;       /* FIXED_SMALL: posterior_log_variance_clipped (:360-363);
                                               FIXED_LARGE: log(append(post_var[1], betas[1:])) (:352-357) */
    const double* sqrt_recip_alphas_cumprod;    /* :184                                              */
    const double* sqrt_recipm1_alphas_cumprod;  /* :185                                              */
    const double* alphas_cumprod;               /* :174                                              */
    const double* alphas_cumprod_prev;          /* :175                                              */
} rgn_schedule;

/* Lifetime. Replaces: CMDM.__init__ (model/cmdm.py:13-111) + model.to(dev()) (sample/cgenerate.py:82). */
RGN_API int rgn_create(const rgn_config* cfg, rgn_handle* out);
RGN_API int rgn_destroy(rgn_handle h);
/* Text of the most recent error on `h` (or of the last failed rgn_create when h == NULL). */
RGN_API const char* rgn_last_error(rgn_handle h);

/* Checkpoint ingestion, keyed by the reference's state_dict names. Replaces
 * load_model_wo_clip / nn.Module.load_state_dict(strict=False) (utils/model_util.py:5-8):
 * an unexpected key -> RGN_ERR_BAD_KEY; keys starting with "clip_model." are accepted and ignored.
 * `host` is caller-owned fp32 host memory, copied during the call. */
RGN_API int rgn_load_weight(rgn_handle h, const char* ref_key, const float* host, const int64_t* shape, int32_t ndim);
/* Checks that no required key is missing (model_util.py:8), folds/packs weights for the kernels and
 * uploads them. Must be called once before any compute entry point. */
RGN_API int rgn_finalize_weights(rgn_handle h);
/* One flat device buffer holds every packed weight so multi-GPU start-up is ONE RCCL broadcast
 * (replaces utils/dist_util.py:54-83 sync_params / load_state_dict). */
RGN_API int rgn_weight_blob(rgn_handle h, void** dev_ptr, uint64_t* nbytes);

/* Replaces SpacedDiffusion.__init__ + _WrappedModel timestep mapping (diffusion/respace.py:64-129)
 * and the per-step _extract_into_tensor gathers (gaussian_diffusion.py:1604-1617). */
RGN_API int rgn_set_schedule(rgn_handle h, const rgn_schedule* s);

/* Binds model_kwargs['y'] for subsequent denoise/sample calls (data_loaders/tensors.py:57-94):
 * cmotion_dev fp32 [B,njoints,nfeats,T]; action_dev int64 [B] (y['action'][:,0]) or NULL;
 * text_feat_dev fp32 [B,clip_dim] = CLIP text features (encode_text output, cmdm.py:153-166) or NULL;
 * scale_dev fp32 [B] (y['scale'], cfg_sampler.py:31) or NULL. Hoists cmo_process(cmotion) and its
 * fuse_process half out of the step loop (model/cmdm.py:202,207-211). Inputs are read, never written. */
RGN_API int rgn_set_condition(rgn_handle h, int32_t B, const float* cmotion_dev, const int64_t* action_dev,
                      const float* text_feat_dev, const float* scale_dev, void* stream);

/* One denoiser evaluation: out = CMDM.forward(x, t, y) (model/cmdm.py:173-252), or with
 * RGN_FLAG_GUIDED the ClassifierFreeSampleModel combination (model/cfg_sampler.py:24-31).
 * t_dev: int64 [B] ORIGINAL timestep indices (what _WrappedModel passes on, respace.py:124-129). */
RGN_API int rgn_denoise(rgn_handle h, const float* x_dev, const int64_t* t_dev, int32_t flags, float* out_dev, void* stream);

/* Steps i = first_index, first_index-1, ..., first_index-count+1 of the sampling loop
 * (p_sample_loop_progressive gaussian_diffusion.py:711-742 / ddim :979-1005) applied in place to
 * x_dev [B,njoints,nfeats,T]:  x <- sampler(x, model(x, map[i]), noise_i).
 *   noise_dev : fp32 [count, B,njoints,nfeats,T], the per-step draws th.randn_like(x)
 *               (gaussian_diffusion.py:544,785) in loop order; NULL -> on-device Philox4x32-10
 *               keyed by (seed, global sample index = sample_offset + b, step i, element).
 *   x0_dev    : optional fp32 output, pred_xstart of the LAST executed step (NULL to skip).
 *   use_graph : replay captured hipGraphs (10 loop iterations each) instead of launching kernels one by one. Honoured by the
 *               throughput kernels; the small-batch engine (<= 640 token rows) launches eagerly either way, which is faster
 *               there (one chain of short kernels; REGENNET_SB_GRAPH=1 forces graphs).
 *   clip_denoised : clamp pred_xstart to [-1,1] (process_xstart, gaussian_diffusion.py:366-372). */
RGN_API int rgn_sample_range(rgn_handle h, int32_t sampler, int32_t guided, float eta, float* x_dev,
                     const float* noise_dev, uint64_t seed, uint64_t sample_offset,
                     int32_t first_index, int32_t count, float* x0_dev, int32_t use_graph,
                     int32_t clip_denoised, void* stream);

/* RGN_PREC_BF16_X3TAIL only: the last `tail_steps` loop indices (i < tail_steps) of every sampling loop run
 * split-bf16, all earlier ones plain bf16. -1 restores the default (max(5, ceil(S/200)) for models of >= 8 layers, max(8, ceil(S/100)) * 8/num_layers for shallower ones, at most S); 0 = plain bf16 throughout;
 * >= S = split-bf16 throughout. Why it is safe: p_sample scales the denoiser output by posterior_mean_coef1[t]
 * (gaussian_diffusion.py:265-276; 0.0016 at t=999, -> 1 at t=0), so early-step rounding is contracted away. */
RGN_API int rgn_set_x3_tail(rgn_handle h, int32_t tail_steps);

/* RGN_PREC_BF16_X3TAIL only, and only where the plain phase runs as the one-kernel decoder stack (k_layers<true>: <= 64 tokens, d = 512, >= 64
 * motions per call): the `steps` plain loop indices right in front of the split-bf16 tail (tail <= i < tail + steps) use IEEE fp16 MFMA operands
 * (11 mantissa bits) instead of bf16 (8) - weights, activation images, q / k / v / p; accumulation, LayerNorm, softmax and the sampler update stay
 * fp32. -1 restores the default (8, or REGENNET_F16_STEPS); 0 = no fp16 phase (then the default split-bf16 tail is the bf16 rule's again). With an
 * fp16 phase the default tail is 2 steps instead of 5 (3 for schedules of <= 10 steps): rgn_set_x3_tail overrides either. fp16 has a 5-bit exponent:
 * rgn_finalize_weights refuses (RGN_ERR_UNSUPPORTED, key named) a checkpoint with a weight of magnitude >= 6e4 unless the "BULK_F16" option is 0.
 * Reference: diffusion/gaussian_diffusion.py:508-560 (why only the last steps' arithmetic reaches the output), model/cmdm.py:227. */
RGN_API int rgn_set_f16_steps(rgn_handle h, int32_t steps);
/* The precision plan rgn_sample_range will follow for B motions on the bound schedule (rgn_set_schedule): loop indices i < *x3_tail run split-bf16,
 * x3_tail <= i < x3_tail + *f16_steps plain fp16 operands, the rest plain bf16 (RGN_PREC_BF16X3: x3_tail = S; other modes: 0 / 0). */
RGN_API int rgn_precision_plan(rgn_handle h, int32_t B, int32_t guided, int32_t* f16_steps, int32_t* x3_tail);

/* const_noise of p_sample (gaussian_diffusion.py:544-547): every motion of the batch receives the per-step draw of the
 * batch's motion 0 (tape entry [k, 0] / the Philox stream of sample_offset + 0); x_T is not affected (:706). Holds for the
 * following rgn_sample_range calls until cleared. */
RGN_API int rgn_set_const_noise(rgn_handle h, int32_t on);

/* Evaluations of at most `rows` token rows (motions x tokens, doubled under guidance) run the small-batch engine:
 * column-split GEMMs that spread one row tile over 16-48 workgroups (rgn_sb.hip; d = 512 models, bf16 modes), the
 * latency-bound regime of the reference CLI's own default batch (sample/cgenerate.py:109-135, BASELINE configs[0]).
 * Larger evaluations run the row-complete throughput kernels. -1 restores the default (640, or REGENNET_SB_ROWS);
 * 0 switches the small-batch engine off. Results of the two engines agree within the precision mode's error (both are
 * checked against the same goldens), not bit for bit. */
RGN_API int rgn_set_small_batch_rows(rgn_handle h, int32_t rows);

/* Kernel-selection switches of ONE handle, set between rgn_create and rgn_finalize_weights (afterwards: RGN_ERR_STATE, except "LAYERS_GUIDED" - a dispatch rule,
 * not a packing decision: 0 an evaluation per workgroup and step | 1 a motion per workgroup when 2 B > #CUs (default) | 2 always | -1 the default again): the names of the
 * REGENNET_<KEY> environment variables without the prefix - "LAYERS" (0: kernel per stage instead of the one-kernel decoder stack), "LAYERS_STEPS",
 * "LAYERS_GUIDED", "LAYERS_MIN_B", "LAYERS_MIN_TQ", "NO_STEP_FUSION", "NO_MLP", "MLP_X3", "NO_ROWGEMM", "NO_FUSED_QKV", "NO_QKV_RS", "QKV_X3_DMA", "NO_QKV_LONG",
 * "SB_ROWS", "SB_FUSED_ATTN", "SB_GRAPH", "STREAMS", "GRAPH_STEPS", "BIG_TILE_ROWS", "BULK_RESID_LO", "STEP_NO_QUADS", "BULK_F16" (0: no fp16
 * weight planes, no fp16 phase), "F16_STEPS"; an unknown name is
 * RGN_ERR_BAD_KEY. A handle's option takes precedence over the environment, which remains the process-wide default (tools, A/B runs): tests and
 * library users address one engine without touching global state. Every selectable form meets the same parity bound. */
RGN_API int rgn_set_option(rgn_handle h, const char* key, int32_t value);

/* Evaluations of at least `samples` samples of <= 64 tokens (motions, doubled under guidance) run the one-kernel decoder stack
 * (rgn_layers.hip: one workgroup per sample, whole runs of sampler steps per launch); smaller ones the kernel-per-stage chain, which
 * spreads a small batch over more CUs. -1 restores the default (64, or REGENNET_LAYERS_MIN_B). The two forms differ by bf16 roundings
 * in the plain-bf16 phase (both inside the parity bound): callers that compare a motion drawn alone with its row of a batch - the
 * precision-schedule calibration, bench.py's row check, the row-independence tests - select the batch's form with this knob. */
RGN_API int rgn_set_layers_min_b(rgn_handle h, int32_t samples);

/* Fills x_dev [B,njoints,nfeats,T] with N(0,1) from the same Philox stream (x_T, gaussian_diffusion.py:706). */
RGN_API int rgn_randn(rgn_handle h, float* x_dev, int32_t B, uint64_t seed, uint64_t sample_offset, void* stream);

/* The noise rgn_sample_range adds at loop index `loop_index` (th.randn_like(x), gaussian_diffusion.py:544 / :792), for
 * callers that run the reference's own per-step loop (p_sample / ddim_sample around rgn_denoise: inpainting masks,
 * y['uncond']) and want the SAME draws as the fused loop for a given (seed, sample_offset): [B,njoints,nfeats,T],
 * value (motion b, feature, frame) = Philox(seed; sample_offset + b, loop_index, feature * 4096 + frame).
 * loop_index = -1 is rgn_randn's x_T draw; loop_index < -1: RGN_ERR_INVALID_ARG. */
RGN_API int rgn_randn_step(rgn_handle h, float* x_dev, int32_t B, uint64_t seed, uint64_t sample_offset, int32_t loop_index, void* stream);

/* Post-processing rows next to the path:
 * rot6d -> rotation matrices, Gram-Schmidt (utils/rotation_conversions.py:513-534): d6 [n,6] -> [n,3,3] */
RGN_API int rgn_rot6d_to_matrix(rgn_handle h, const float* d6_dev, float* mat_dev, int64_t n, void* stream);
/* scipy.ndimage.gaussian_filter1d(x, sigma, axis=-1, mode='reflect') on device (sample/cgenerate.py:142):
 * x [rows, T] -> out [rows, T] */
RGN_API int rgn_gaussian_filter1d(rgn_handle h, const float* x_dev, float* out_dev, int64_t rows, int32_t T,
                          float sigma, void* stream);

/* Introspection for bench/profiling: name and accumulated HIP-event time (ms) + launch count of the
 * internal kernel classes since the last reset; timing is only collected when enabled. */
RGN_API int rgn_profile_enable(rgn_handle h, int32_t on);
RGN_API int rgn_profile_query(rgn_handle h, int32_t idx, const char** name, double* total_ms, int64_t* launches);
/* The engine's own plan for ONE denoiser evaluation of a sampling loop over B motions (2 B rows when `guided`), in the plain-bf16
 * phase of the precision schedule (split_phase = 0) or its split-bf16 tail (1): for kernel class idx (the classes of rgn_profile_query)
 * the concrete kernel, launches per evaluation (single kernel chain; 0 for k_layers<true>, which is ONE launch per run of steps),
 * the algorithmic FLOPs those launches carry (2 x MAC, full T x T scores) and, for the one-kernel forms, the weight
 * fragment bytes their workgroups stream from L2. Filled by the code that dispatches (plan_eval in rgn_plan.cpp), so a benchmark prices
 * exactly what the engine launches. */
RGN_API int rgn_plan_query(rgn_handle h, int32_t B, int32_t guided, int32_t split_phase, int32_t idx, const char** name,
                           const char** kernel, double* launches_per_eval, double* algo_flops_per_eval, double* l2_bytes_per_eval);
/* Time (ms) one event pair measures around a one-thread no-op kernel on this handle's stream: the dispatch + event
 * latency every bracket of rgn_profile_query carries on top of the kernel itself (calibrated at the first enable). */
RGN_API int rgn_profile_bracket_overhead(rgn_handle h, double* ms);

/* ---- Evaluation harness next to the sampler: the ST-GCN feature extractor / action classifier ----
 * Replaces eval/a2m/recognition/models/stgcn.py:28-123 (STGCN) as used by eval/a2m/stgcn/evaluate.py:9-45: the features
 * feed FID / diversity / multimodality, yhat the accuracy. Checkpoint keys are the reference's state_dict names
 * ('A', 'data_bn.*', 'st_gcn_networks.<i>.{gcn.conv,tcn.0,tcn.2,tcn.3,residual.0,residual.1}.*', 'edge_importance.<i>',
 * 'fcn.*'); '*.num_batches_tracked' entries are accepted and ignored. */
typedef struct rgn_stgcn_ctx* rgn_stgcn_handle;
typedef struct {
    int32_t in_channels;    /* nfeats of batch['output'] = C * num_person (evaluate.py:15; 12 for two persons in rot6d) */
    int32_t num_class;      /* evaluate.py:16 */
    int32_t num_person;     /* evaluate.py:17 */
    int32_t num_nodes;      /* V: graph nodes = joints (56 for the 'smplx' layout, stgcnutils/graph.py:81-82) */
    int32_t num_frames;     /* T of batch['output'] */
    int32_t max_batch;
    int32_t device;
} rgn_stgcn_config;
RGN_API int rgn_stgcn_create(const rgn_stgcn_config* cfg, rgn_stgcn_handle* out);
RGN_API int rgn_stgcn_destroy(rgn_stgcn_handle h);
RGN_API const char* rgn_stgcn_last_error(rgn_stgcn_handle h);
/* host fp32 copies of the checkpoint tensors (model.load_state_dict, evaluate.py:24-25) */
RGN_API int rgn_stgcn_load_weight(rgn_stgcn_handle h, const char* ref_key, const float* host, const int64_t* shape, int32_t ndim);
/* folds every BatchNorm (eval mode) and the edge importance into the convolutions; RGN_ERR_MISSING_KEY names what is absent */
RGN_API int rgn_stgcn_finalize(rgn_stgcn_handle h);
/* Kernel-selection switches of ONE recogniser handle (like rgn_set_option; any time before a forward): "SG_NO_WINDOW" (row-shifted GEMMs instead of the
 * LDS-window temporal convolutions), "SG_NO_GCN_FUSE" (aggregation and 1x1 convolution as two launches), "SG_NO_TAIL_FUSE" (k_sg_post for every block), "SG_NO_POLY_TAIL" (... for the two blocks with polyphase output),
 * "SG_NO_S2_WINDOW" (stride-2 blocks as row-shifted GEMM + shortcut GEMM), "SG_TCONV_SMALL" (256-row tiles), "SG_GCN_BN" (widest aggregation tile: 64 |
 * 128 | 256), "SG_GCN_STEP32" (64-wide aggregation tiles: one barrier per 32-deep k-block), "SG_NO_BLOCK0_FUSE" (the first block's graph convolution as aggregation + split GEMM instead of one fp32 kernel). A handle's option takes precedence over REGENNET_<KEY> in the environment; unknown names are RGN_ERR_BAD_KEY. Every form meets the same
 * parity bound (tests/test_eval_gpu.py runs each against the reference's outputs).
 * "SG_F16" (0 | 1, default 0) is not a kernel form but the ARITHMETIC: blocks 1-9 and block 0's temporal convolution on single IEEE fp16 operand planes, one MFMA per
 * product instead of the split-bf16 three (features within 1.5e-3 of the largest feature - measured 4e-4 - instead of 1e-4 / 5e-6; about twice the speed). It exists
 * for the fused kernels only: rgn_stgcn_forward returns RGN_ERR_UNSUPPORTED, with the reason in rgn_stgcn_last_error, for a graph / shape / SG_NO_* selection they do
 * not cover or a checkpoint with a folded weight of magnitude >= 6e4 (key named) - never a silent change of arithmetic. */
RGN_API int rgn_stgcn_set_option(rgn_stgcn_handle h, const char* key, int32_t value);
/* STGCN.forward (stgcn.py:76-123): output_dev fp32 [N, num_nodes, in_channels, T] (batch['output']) ->
 * features_dev fp32 [N, 256] (batch['features'], nullable) and yhat_dev fp32 [N, num_class] (batch['yhat'], nullable) */
RGN_API int rgn_stgcn_forward(rgn_stgcn_handle h, int32_t N, const float* output_dev, float* features_dev, float* yhat_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* REGENNET_HIP_H */
