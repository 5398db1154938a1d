"""Batched `auto_regressive` generation.

Reference: eval/a2m/stgcn_eval.py:50-67 (`NewDataloader(..., auto_regressive=True)`). For every frame index f the reference
reveals the actor's frames 0..f (later frames zero), runs the WHOLE sampler with fresh noise and keeps frame f of the
result: T sequential sampler calls of batch B per evaluation batch (60 x 1000 denoiser evaluations on NTU120-AS).

The T runs are independent of each other (each draws its own noise, none consumes another's output), so here they are
the samples of ONE larger batch: sample (f, b) carries the condition of b with the actor motion masked after frame f, and
`frames_per_call` frames (B * frames_per_call samples) go through one `p_sample_loop` call, i.e. through the engine's
batched kernels and a single captured hipGraph. With the on-device Philox stream, sample (f, b) is keyed by the global
index f * B + b, so the result does not depend on how the frames are grouped into calls.

Sequence truncation: the decoder is causal (cmdm.py:168-171,220-227) and only frame f of run f is kept, which depends on
tokens 0..f alone at every diffusion step. A call that covers the frames [f0, f1) therefore samples sequences of f1 tokens
(rounded up to a multiple of 16, which bounds the number of per-length engines) instead of T (the model object is length-agnostic, like the reference's): about half the work over a whole evaluation. Noise
is keyed by (sample, step, feature, frame), independent of the sequence length, so truncation changes no noise value; the kept
frames then agree with the untruncated run to within the precision mode's rounding (bit for bit on the small-batch engine with
uniform split-bf16 arithmetic, which is what the tests pin; a different sequence length can select different kernels).
"""
import torch as th

# tensors in y that are per-sample (leading dim B) and must follow the (frame, sample) flattening
_PER_SAMPLE = ("cmotion", "action", "action_cond", "text_features", "scale", "lengths", "mask", "trans_mask")


def _expand_y(y, B, f0, f1, cm_full, T_call=None):
    """model_kwargs['y'] for the samples (f, b), f in [f0, f1): index (f - f0) * B + b; sequences cut to T_call frames."""
    n = f1 - f0
    T = cm_full.shape[-1]
    T_call = T if T_call is None else T_call
    out = {}
    for k, v in y.items():
        if k == "cmotion":
            continue
        if th.is_tensor(v) and v.dim() >= 1 and v.shape[0] == B:
            if k in ("mask", "trans_mask") and v.shape[-1] == T:
                v = v[..., :T_call]
            if k == "lengths":
                v = v.clamp(max=T_call)
            out[k] = v.repeat((n,) + (1,) * (v.dim() - 1))
        elif isinstance(v, (list, tuple)) and len(v) == B:
            out[k] = list(v) * n
        else:
            out[k] = v
    # frame f sees the actor's frames 0..f; everything after is zero (stgcn_eval.py:52,58)
    frames = th.arange(T, device=cm_full.device)
    keep = (frames[None, :] <= th.arange(f0, f1, device=cm_full.device)[:, None]).to(cm_full.dtype)   # [n, T]
    cm = (cm_full[None] * keep[:, None, None, None, :])[..., :T_call]                               # [n, B, V, C, T_call]
    out["cmotion"] = cm.reshape((n * B,) + tuple(cm_full.shape[1:3]) + (T_call,)).contiguous()
    return out


def sample_auto_regressive(sample_fn, model, shape, model_kwargs, setting="cmdm", frames_per_call=None,
                           clip_denoised=False, noise_tapes=None, seed=None, max_samples_per_call=256, truncate=True,
                           **sample_kw):
    """Return the reference's `batch['output']` of the auto-regressive branch.

    sample_fn       diffusion.p_sample_loop or diffusion.ddim_sample_loop (regennet_amd.diffusion).
    shape           (B, njoints, nfeats, T) as passed to the reference's sample_fn (motions.shape).
    model_kwargs    {'y': {...}} with y['cmotion'] [B, V, C, T] the full actor motion.
    setting         'cmdm': output = cat(actor, reactor) on axis 2 -> [B, V, 2C, T]; otherwise the reactor only.
    frames_per_call frames batched into one sampler call (default: as many as keep B * frames <= max_samples_per_call).
    truncate        sample only the first f1 tokens in the call that covers frames [f0, f1) (see the module docstring).
    noise_tapes     optional sequence of T tapes [S+1, B, V, C, T] (run f consumes tape f; for parity tests),
    seed            otherwise the Philox seed (None: drawn from torch's generator, like the sampler itself).
    Remaining keyword arguments go to sample_fn (eta, skip_timesteps, use_graph, ...).
    """
    y = model_kwargs["y"]
    cm_full = y["cmotion"]
    B, V, C, T = cm_full.shape
    assert tuple(shape) == (B, V, C, T), f"shape {tuple(shape)} != cmotion {tuple(cm_full.shape)}"
    if frames_per_call is None:
        frames_per_call = max(1, max_samples_per_call // B)
    frames_per_call = max(1, min(int(frames_per_call), T))
    if seed is None and noise_tapes is None:
        seed = int(th.randint(0, 2 ** 62, (1,), dtype=th.int64).item())
    output = None
    for f0 in range(0, T, frames_per_call):
        f1 = min(T, f0 + frames_per_call)
        n = f1 - f0
        # lengths in steps of 16 (extra tokens are harmless under the causal mask): at most ceil(T / 16) distinct engines
        # however the frames are grouped, below the model's per-length engine cache (CMDM.MAX_ENGINES)
        Tc = min(T, -(-f1 // 16) * 16) if truncate else T
        yy = _expand_y(y, B, f0, f1, cm_full, Tc)
        kw = dict(sample_kw)
        if noise_tapes is not None:
            kw["noise_tape"] = th.cat([th.as_tensor(noise_tapes[f])[..., :Tc] for f in range(f0, f1)], dim=1).contiguous()   # batch axis
        else:
            kw.update(seed=seed, sample_offset=f0 * B + int(sample_kw.get("sample_offset", 0)))
        sample = sample_fn(model, (n * B, V, C, Tc), clip_denoised=clip_denoised, model_kwargs={"y": yy}, **kw)
        sample = sample.reshape(n, B, V, C, Tc)
        if output is None:
            output = th.zeros((B, V, C * 2 if setting == "cmdm" else C, T), device=sample.device, dtype=sample.dtype)
        idx = th.arange(f0, f1, device=sample.device)
        picked = sample[th.arange(n, device=sample.device), :, :, :, idx]          # [n, B, V, C]: frame f of run f
        if setting == "cmdm":
            output[:, :, C:, f0:f1] = picked.permute(1, 2, 3, 0)
        else:
            output[:, :, :, f0:f1] = picked.permute(1, 2, 3, 0)
    if setting == "cmdm":
        output[:, :, :C, :] = cm_full.to(device=output.device, dtype=output.dtype)   # frame f of the masked actor motion = the actor's frame f
    return output
