// The C-ABI of libregennet_hip.so (include/regennet_hip.h): argument and call-order checks, the exception guard every entry point runs in,
// and the entry points that are a few lines of host code. Checkpoint packing is rgn_pack.cpp, planning and dispatch rgn_plan.cpp.
#include "rgn_host.h"

using namespace rgnh;

namespace rgnh {
const char* const kclass_names[KC_COUNT] = {"gemm_mfma", "attention", "layernorm", "embed", "update", "misc", "qkv_attn", "rowgemm_ln", "rowgemm_act", "mlp", "sb_gemm", "step_fused", "layers", "steps_fused"};

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
}  // namespace rgnh

namespace {

thread_local std::string g_create_error;   // text of the calling thread's last failed rgn_create

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
        return finalize_weights(h);
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
        return sample_range(h, sampler, guided, eta, x, noise, seed, sample_offset, first_index, count, x0_out, use_graph, clip_denoised, stream);
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
        h->f16_steps = steps < 0 ? h->f16_steps_default : steps;
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
        if (h->finalized && strcmp(key, "LAYERS_GUIDED") == 0) {   // (a dispatch rule, not a packing decision: may change between calls; captured graphs hold the old form)
            const int v = value < 0 ? h->layers_guided_default : (value > 2 ? 2 : value);
            if (v != h->layers_guided) {
                if (h->stream) RGN_HIP(h, hipStreamSynchronize(h->stream));
                for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
                h->graphs.clear();
                h->layers_guided = v;
            }
            return RGN_OK;
        }
        if (h->finalized) return h->fail(RGN_ERR_STATE, "rgn_set_option: the switches select kernels when the weights are packed - set them before rgn_finalize_weights");
        static const char* known[] = {"NO_FUSED_QKV", "BIG_TILE_ROWS", "NO_ROWGEMM", "NO_MLP", "MLP_X3", "NO_QKV_RS", "NO_STEP_FUSION", "LAYERS_MIN_TQ", "LAYERS", "LAYERS_STEPS",
                                      "LAYERS_MIN_B", "LAYERS_GUIDED", "STEP_NO_QUADS", "NO_QKV_LONG", "SB_FUSED_ATTN", "SB_ROWS", "BULK_RESID_LO", "GRAPH_STEPS", "STREAMS", "SB_GRAPH",
                                      "BULK_F16", "F16_STEPS", "QKV_X3_DMA"};
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
        return plan_query(h, B, guided, split_phase, idx, name, kernel, launches_per_eval, algo_flops_per_eval, l2_bytes_per_eval);
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
