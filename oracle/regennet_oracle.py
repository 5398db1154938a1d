"""ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.

CPU restatement (PyTorch CPU tensor ops, fp32; schedule maths in NumPy fp64) of the ReGenNet
diffusion-sampling hot path, written from the algorithm description in SURVEY.md §3/§8a with the
reference file:line each function follows. Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; regennet_amd/ never does.

Parity pinning: the reference ships NO tests/golden vectors for this path, so this
oracle is pinned against outputs of the reference itself, run in the build container by
tests/golden/make_golden.py (fixtures committed under tests/golden/*.npz) and against the
known-answer values of the schedule tables recorded in SURVEY.md §8a (a1/a2/a3).

Third-party arithmetic restated here: torch.nn.TransformerDecoderLayer (post-norm, eps 1e-5,
exact-erf GELU), nn.MultiheadAttention (packed in_proj, scale 1/sqrt(dh)), nn.Linear, nn.LayerNorm,
nn.SiLU — PyTorch, pinned by the reference at pytorch=1.7.1 (environment.yml:88) /
1.12.0 (docker/Dockerfile:1); constructed at model/cmdm.py:75-81 and called at :227.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# a1/a2: schedules and tables            diffusion/gaussian_diffusion.py:21-65, :172-209
# --------------------------------------------------------------------------------------
def get_named_beta_schedule(name, n, scale_betas=1.0):
    if name == "linear":                                   # gaussian_diffusion.py:30-38
        scale = scale_betas * 1000 / n
        return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)
    if name == "cosine":                                   # :39-43 -> betas_for_alpha_bar :48-65
        ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1 - ab((i + 1) / n) / ab(i / n), 0.999) for i in range(n)])
    raise NotImplementedError(f"unknown beta schedule: {name}")


def diffusion_tables(betas):
    """All per-timestep fp64 tables of GaussianDiffusion.__init__ (gaussian_diffusion.py:172-209)."""
    betas = np.array(betas, dtype=np.float64)
    assert betas.ndim == 1 and (betas > 0).all() and (betas <= 1).all()
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    return dict(
        betas=betas,
        alphas_cumprod=ac,
        alphas_cumprod_prev=ac_prev,
        alphas_cumprod_next=np.append(ac[1:], 0.0),
        sqrt_alphas_cumprod=np.sqrt(ac),
        sqrt_one_minus_alphas_cumprod=np.sqrt(1.0 - ac),
        sqrt_recip_alphas_cumprod=np.sqrt(1.0 / ac),
        sqrt_recipm1_alphas_cumprod=np.sqrt(1.0 / ac - 1),
        posterior_variance=post_var,
        posterior_log_variance_clipped=np.log(np.append(post_var[1], post_var[1:])),
        posterior_mean_coef1=betas * np.sqrt(ac_prev) / (1.0 - ac),
        posterior_mean_coef2=(1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
    )


# --------------------------------------------------------------------------------------
# a3/a4: timestep respacing                                   diffusion/respace.py:8-87
# --------------------------------------------------------------------------------------
def space_timesteps(num_timesteps, section_counts):
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):              # respace.py:30-38
            want = int(section_counts[len("ddim"):])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == want:
                    return set(range(0, num_timesteps, i))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, cnt in enumerate(section_counts):               # respace.py:44-60
        size = size_per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError(f"cannot divide section of {size} steps into {cnt}")
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return set(steps)


def spaced_schedule(base_betas, use_timesteps):
    """(timestep_map, tables) of SpacedDiffusion.__init__ (respace.py:73-87)."""
    use = set(use_timesteps)
    ac = np.cumprod(1.0 - np.array(base_betas, dtype=np.float64))
    last, new_betas, tmap = 1.0, [], []
    for i, a in enumerate(ac):
        if i in use:
            new_betas.append(1 - a / last)
            last = a
            tmap.append(i)
    return tmap, diffusion_tables(np.array(new_betas))


def make_schedule(noise_schedule="cosine", timestep_respacing="", steps=1000, sigma_small=True):
    """utils/model_util.py:75-117 (steps hard-coded to 1000 at :78). `model_log_variance` is the table p_mean_variance
    picks for FIXED_SMALL / FIXED_LARGE (gaussian_diffusion.py:344-364)."""
    betas = get_named_beta_schedule(noise_schedule, steps, 1.0)
    tmap, tables = spaced_schedule(betas, space_timesteps(steps, timestep_respacing or [steps]))
    if sigma_small:
        tables["model_log_variance"] = tables["posterior_log_variance_clipped"]
    else:
        tables["model_log_variance"] = np.log(np.append(tables["posterior_variance"][1], tables["betas"][1:]))
    return tmap, tables


def _extract(arr, t, shape):
    """_extract_into_tensor (gaussian_diffusion.py:1604-1617): fp64 table -> index -> fp32 -> broadcast."""
    res = torch.from_numpy(arr)[t].float()
    while res.dim() < len(shape):
        res = res[..., None]
    return res.expand(shape)


# --------------------------------------------------------------------------------------
# a12/a13: CMDM.forward, arch='online'                              model/cmdm.py:173-252
# --------------------------------------------------------------------------------------
def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def _mha_self(x, sd, p, nheads, mask):
    """nn.MultiheadAttention self-attention, seq-first x [T,B,d], additive float mask [T,T]."""
    T, B, d = x.shape
    dh = d // nheads
    qkv = F.linear(x, _t(sd, p + "in_proj_weight"), _t(sd, p + "in_proj_bias"))
    q, k, v = qkv.chunk(3, dim=-1)
    q = q.reshape(T, B * nheads, dh).transpose(0, 1)
    k = k.reshape(T, B * nheads, dh).transpose(0, 1)
    v = v.reshape(T, B * nheads, dh).transpose(0, 1)
    s = torch.bmm(q, k.transpose(1, 2)) / math.sqrt(dh) + mask
    o = torch.bmm(torch.softmax(s, dim=-1), v)
    o = o.transpose(0, 1).reshape(T, B, d)
    return F.linear(o, _t(sd, p + "out_proj.weight"), _t(sd, p + "out_proj.bias"))


def _mha_cross_one_token(mem, sd, p):
    """Cross-attention onto a 1-token memory [1,B,d]: softmax over one key == 1, so the output is
    out_proj(v_proj(mem)) for every query position."""
    d = mem.shape[-1]
    w, b = _t(sd, p + "in_proj_weight"), _t(sd, p + "in_proj_bias")
    v = F.linear(mem, w[2 * d:], b[2 * d:])
    return F.linear(v, _t(sd, p + "out_proj.weight"), _t(sd, p + "out_proj.bias"))  # [1,B,d]


def cmdm_forward(sd, cfg, x, timesteps, y):
    """x [B,J,F,T] fp32, timesteps [B] int64 (ORIGINAL 0..999 indices), y dict -> x0_hat [B,J,F,T]."""
    B, J, Fe, T = x.shape
    d, H, L = cfg["latent_dim"], cfg["num_heads"], cfg["layers"]
    pe = _t(sd, "sequence_pos_encoder.pe")                                   # [5000,1,d]
    # TimestepEmbedder.forward cmdm.py:297-298
    emb = F.linear(F.silu(F.linear(pe[timesteps], _t(sd, "embed_timestep.time_embed.0.weight"),
                                   _t(sd, "embed_timestep.time_embed.0.bias"))),
                   _t(sd, "embed_timestep.time_embed.2.weight"), _t(sd, "embed_timestep.time_embed.2.bias"))
    emb = emb.permute(1, 0, 2)                                               # [1,B,d]
    force_mask = y.get("uncond", False)                                      # cmdm.py:181
    if "text" in cfg["cond_mode"]:                                           # cmdm.py:182-184 (CLIP features supplied)
        enc = y["text_features"]
        enc = torch.zeros_like(enc) if force_mask else enc
        emb = emb + F.linear(enc, _t(sd, "embed_text.weight"), _t(sd, "embed_text.bias"))
    if "action" in cfg["cond_mode"]:                                         # cmdm.py:185-187, :363-365
        a = _t(sd, "embed_action.action_embedding")[y["action"][:, 0].long()]
        emb = emb + (torch.zeros_like(a) if force_mask else a)
    # InputProcess cmdm.py:311-317 (x2), fuse cmdm.py:207-211
    xs = x.permute(3, 0, 1, 2).reshape(T, B, J * Fe)
    cs = y["cmotion"].permute(3, 0, 1, 2).reshape(T, B, J * Fe)
    xs = F.linear(xs, _t(sd, "input_process.poseEmbedding.weight"), _t(sd, "input_process.poseEmbedding.bias"))
    cs = F.linear(cs, _t(sd, "cmo_process.poseEmbedding.weight"), _t(sd, "cmo_process.poseEmbedding.bias"))
    if cfg["cm_mode"] == "add":
        xseq = xs + cs
    else:
        xseq = F.linear(torch.cat((xs, cs), dim=-1), _t(sd, "fuse_process.weight"), _t(sd, "fuse_process.bias"))
    etd = cfg.get("emb_trans_dec", False)
    if etd:                                                                  # cmdm.py:212-213
        xseq = torch.cat((emb, xseq), dim=0)
    if not cfg.get("wo_pos_emb", False):                                     # cmdm.py:217-218, :278-281
        xseq = xseq + pe[: xseq.shape[0]]
    S = xseq.shape[0]
    mask = torch.full((S, S), float("-inf")).triu(1)                         # cmdm.py:168-171
    for l in range(L):                                                       # post-norm decoder layer
        p = f"seqTransDecoder.layers.{l}."
        xseq = F.layer_norm(xseq + _mha_self(xseq, sd, p + "self_attn.", H, mask), (d,),
                            _t(sd, p + "norm1.weight"), _t(sd, p + "norm1.bias"), 1e-5)
        xseq = F.layer_norm(xseq + _mha_cross_one_token(emb, sd, p + "multihead_attn."), (d,),
                            _t(sd, p + "norm2.weight"), _t(sd, p + "norm2.bias"), 1e-5)
        h = F.gelu(F.linear(xseq, _t(sd, p + "linear1.weight"), _t(sd, p + "linear1.bias")))
        xseq = F.layer_norm(xseq + F.linear(h, _t(sd, p + "linear2.weight"), _t(sd, p + "linear2.bias")), (d,),
                            _t(sd, p + "norm3.weight"), _t(sd, p + "norm3.bias"), 1e-5)
    if etd:
        xseq = xseq[1:]                                                      # cmdm.py:225
    out = F.linear(xseq, _t(sd, "output_process.poseFinal.weight"), _t(sd, "output_process.poseFinal.bias"))
    return out.reshape(T, B, J, Fe).permute(1, 2, 3, 0)                      # cmdm.py:353-354


def cfg_forward(sd, cfg, x, timesteps, y):
    """ClassifierFreeSampleModel.forward (model/cfg_sampler.py:24-31)."""
    assert cfg["cond_mode"] in ("text", "action") and cfg["cond_mask_prob"] > 0
    out = cmdm_forward(sd, cfg, x, timesteps, y)
    yu = dict(y)
    yu["uncond"] = True
    out_u = cmdm_forward(sd, cfg, x, timesteps, yu)
    return out_u + y["scale"].view(-1, 1, 1, 1) * (out - out_u)


# --------------------------------------------------------------------------------------
# a5-a10: sampling loops                      gaussian_diffusion.py:289-400,508-560,610-794,891-1005
# --------------------------------------------------------------------------------------
def sample_loop(sd, cfg, schedule, tape, y, mode="ddpm", guided=False, eta=0.0, trace=None, clip_denoised=False,
                skip_timesteps=0, init_image=None):
    """Run p_sample_loop (mode='ddpm') or ddim_sample_loop (mode='ddim') with an injected noise tape.

    tape[0] = x_T, tape[k] = k-th per-step draw (drawn every step, gaussian_diffusion.py:544,785).
    `trace`, if a dict, receives per-step 'x0' and 'x' lists. Returns the final sample [B,J,F,T].
    """
    tmap, tb = schedule
    tmap_t = torch.tensor(tmap, dtype=torch.long)
    S = len(tmap)
    img = torch.as_tensor(tape[0]).clone()
    B = img.shape[0]
    fwd = cfg_forward if guided else cmdm_forward
    k = 1
    if skip_timesteps and init_image is None:                                # gaussian_diffusion.py:708-709
        init_image = torch.zeros_like(img)
    first = S - 1 - skip_timesteps
    if init_image is not None:                                               # :713-715 -> q_sample :248-266 with noise=img
        t0 = torch.full((B,), first, dtype=torch.long)
        img = _extract(tb["sqrt_alphas_cumprod"], t0, img.shape) * torch.as_tensor(init_image) + \
            _extract(tb["sqrt_one_minus_alphas_cumprod"], t0, img.shape) * img
    with torch.no_grad():
        for i in range(first, -1, -1):
            t = torch.tensor([i] * B)                                        # gaussian_diffusion.py:724
            x0 = fwd(sd, cfg, img, tmap_t[t], y)                             # respace.py:124-129; START_X :381
            if clip_denoised:
                x0 = x0.clamp(-1, 1)                                         # process_xstart :366-372
            noise = torch.as_tensor(tape[k]); k += 1
            nz = (t != 0).float().view(-1, 1, 1, 1)
            if mode == "ddpm":                                               # p_sample :508-560; FIXED_SMALL :344-364
                mean = _extract(tb["posterior_mean_coef1"], t, img.shape) * x0 + \
                       _extract(tb["posterior_mean_coef2"], t, img.shape) * img
                logvar = _extract(tb.get("model_log_variance", tb["posterior_log_variance_clipped"]), t, img.shape)
                img = mean + nz * torch.exp(0.5 * logvar) * noise
            else:                                                            # ddim_sample :744-794
                eps = (_extract(tb["sqrt_recip_alphas_cumprod"], t, img.shape) * img - x0) / \
                      _extract(tb["sqrt_recipm1_alphas_cumprod"], t, img.shape)
                ab = _extract(tb["alphas_cumprod"], t, img.shape)
                abp = _extract(tb["alphas_cumprod_prev"], t, img.shape)
                sigma = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
                mean = x0 * torch.sqrt(abp) + torch.sqrt(1 - abp - sigma ** 2) * eps
                img = mean + nz * sigma * noise
            if trace is not None:
                trace.setdefault("x0", []).append(x0.clone())
                trace.setdefault("x", []).append(img.clone())
    return img


# --------------------------------------------------------------------------------------
# next-3 row: auto_regressive generation           eval/a2m/stgcn_eval.py:50-67
# --------------------------------------------------------------------------------------
def sample_auto_regressive(sd, cfg, schedule, tapes, y, setting="cmdm", mode="ddpm"):
    """Literal restatement of the reference's frame loop: for frame f the actor's frames 0..f are revealed (the rest
    stay zero), one complete sampler run is made with fresh noise (tapes[f]) and only frame f of the result is kept.
    Returns `output` [B, J, 2F, T] for setting 'cmdm' (actor rows then reactor rows, stgcn_eval.py:62-63), else [B,J,F,T]."""
    cm_full = torch.as_tensor(y["cmotion"])
    B, J, Fe, T = cm_full.shape
    cm = torch.zeros_like(cm_full)
    out = torch.zeros((B, J, Fe * 2 if setting == "cmdm" else Fe, T))
    for f in range(T):
        cm[:, :, :, f] = cm_full[:, :, :, f]                                 # :58
        yy = dict(y)
        yy["cmotion"] = cm
        sample = sample_loop(sd, cfg, schedule, tapes[f], yy, mode=mode, clip_denoised=False)   # :60
        tmp = torch.cat((cm, sample), dim=2) if setting == "cmdm" else sample    # :61-64
        out[:, :, :, f] = tmp[:, :, :, f]                                    # :65
    return out


# --------------------------------------------------------------------------------------
# next-1 / next-2 rows
# --------------------------------------------------------------------------------------
def rotation_6d_to_matrix(d6):
    """utils/rotation_conversions.py:513-534 (Gram-Schmidt, rows b1,b2,b3)."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = F.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def gaussian_filter1d_lastaxis(x, sigma=1.0, truncate=4.0):
    """scipy.ndimage.gaussian_filter1d(x, sigma, axis=-1) with default mode='reflect'
    (sample/cgenerate.py:142). NumPy restatement: radius=int(truncate*sigma+.5) taps, symmetric
    half-sample reflection (d c b a | a b c d | d c b a)."""
    x = np.asarray(x)
    r = int(truncate * float(sigma) + 0.5)
    k = np.exp(-0.5 / (sigma * sigma) * np.arange(-r, r + 1) ** 2)
    k /= k.sum()
    n = x.shape[-1]
    idx = np.arange(-r, n + r)
    period = 2 * n
    idx = np.mod(idx, period)
    idx = np.where(idx >= n, period - 1 - idx, idx)
    xp = x[..., idx].astype(np.float64)
    out = np.zeros(x.shape, dtype=np.float64)
    for j in range(2 * r + 1):
        out += k[j] * xp[..., j:j + n]
    return out.astype(x.dtype)
