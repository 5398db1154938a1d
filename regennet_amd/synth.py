"""Synthetic checkpoints and inputs for the ReGenNet sampling hot path.

No trained checkpoint, dataset or SMPL-X asset exists in the build environment,
so tests and bench.py use a deterministic synthetic checkpoint emitted with the *reference's*
state_dict key names and shapes (model/cmdm.py:53-105 in the reference), and synthetic actor
motions laid out like `ccollate` produces them (data_loaders/tensors.py:57-94): `cmotion`
[B, 56, 6, T] = 55 SMPL-X joints in rot6d + one translation row [tx,ty,tz,0,0,0]
(data_loaders/a2m/dataset.py:175-181).

Everything here is NumPy (PCG64) so it is reproducible on any box without the reference.
"""
import numpy as np

# Named configurations mirroring BASELINE.json `configs`.
CONFIGS = {
    # NTU120-AS shipped config: README.md:96 + utils/parser_util.py defaults + utils/model_util.py:66-72
    "ntu": dict(dataset="ntu", njoints=56, nfeats=6, num_frames=60, latent_dim=512, ff_size=1024,
                num_heads=4, layers=8, cm_mode="concat", cond_mode="no_cond", cond_mask_prob=0.0,
                num_actions=26),
    "ntu_action": dict(dataset="ntu", njoints=56, nfeats=6, num_frames=60, latent_dim=512, ff_size=1024,
                       num_heads=4, layers=8, cm_mode="concat", cond_mode="action", cond_mask_prob=0.1,
                       num_actions=26),
    "chi3d": dict(dataset="chi3d", njoints=56, nfeats=6, num_frames=150, latent_dim=512, ff_size=1024,
                  num_heads=4, layers=8, cm_mode="concat", cond_mode="action", cond_mask_prob=0.1,
                  num_actions=8),
    "text150": dict(dataset="chi3d", njoints=56, nfeats=6, num_frames=150, latent_dim=512, ff_size=1024,
                    num_heads=4, layers=8, cm_mode="concat", cond_mode="text", cond_mask_prob=0.1,
                    num_actions=1),
    # small configs the oracle / reference finish in milliseconds
    "tiny": dict(dataset="ntu", njoints=5, nfeats=6, num_frames=8, latent_dim=64, ff_size=128,
                 num_heads=4, layers=2, cm_mode="concat", cond_mode="action", cond_mask_prob=0.1,
                 num_actions=3),
    "tiny_add": dict(dataset="ntu", njoints=5, nfeats=6, num_frames=8, latent_dim=64, ff_size=128,
                     num_heads=4, layers=2, cm_mode="add", cond_mode="no_cond", cond_mask_prob=0.0,
                     num_actions=1),
    "tiny_text": dict(dataset="ntu", njoints=5, nfeats=6, num_frames=11, latent_dim=64, ff_size=128,
                      num_heads=2, layers=2, cm_mode="concat", cond_mode="text", cond_mask_prob=0.1,
                      num_actions=1),
}


def get_config(name, **overrides):
    cfg = dict(CONFIGS[name])
    cfg.update(overrides)
    return cfg


def positional_table(d_model, max_len=5000):
    """Sinusoid table [max_len, 1, d] exactly as PositionalEncoding builds it (model/cmdm.py:269-276):
    fp32 arithmetic throughout (torch.arange(...).float(), torch.exp, torch.sin/cos on fp32)."""
    position = np.arange(0, max_len, dtype=np.float32)[:, None]
    div_term = np.exp(np.arange(0, d_model, 2, dtype=np.float32) * np.float32(-np.log(10000.0) / d_model))
    div_term = div_term.astype(np.float32)
    pe = np.zeros((max_len, d_model), dtype=np.float32)
    ang = (position * div_term).astype(np.float32)
    pe[:, 0::2] = np.sin(ang)
    pe[:, 1::2] = np.cos(ang)
    return pe[:, None, :]


def make_state_dict(cfg, seed=0):
    """Deterministic synthetic checkpoint: dict key -> float32 ndarray, reference key names.

    Linear weights ~ N(0, gain^2/fan_in); q/k projections get a larger gain so the causal softmax is
    far from uniform; LayerNorm gamma ~ 1+0.1N, beta ~ 0.1N; biases ~ 0.02..0.1 N.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    d, ff, L = cfg["latent_dim"], cfg["ff_size"], cfg["layers"]
    F = cfg["njoints"] * cfg["nfeats"]
    sd = {}

    def lin(prefix, out_f, in_f, gain=1.0, bias_std=0.05):
        sd[prefix + ".weight"] = (rng.standard_normal((out_f, in_f)) * (gain / np.sqrt(in_f))).astype(np.float32)
        sd[prefix + ".bias"] = (rng.standard_normal(out_f) * bias_std).astype(np.float32)

    lin("input_process.poseEmbedding", d, F)
    lin("cmo_process.poseEmbedding", d, F)
    if cfg["cm_mode"] == "concat":
        lin("fuse_process", d, 2 * d, gain=1.4)
    pe = positional_table(d)
    sd["sequence_pos_encoder.pe"] = pe
    sd["embed_timestep.sequence_pos_encoder.pe"] = pe.copy()
    lin("embed_timestep.time_embed.0", d, d, gain=1.5)
    lin("embed_timestep.time_embed.2", d, d, gain=1.5)
    for l in range(L):
        p = f"seqTransDecoder.layers.{l}."
        w = rng.standard_normal((3 * d, d)) / np.sqrt(d)
        w[: 2 * d] *= 2.5  # q,k gain -> peaky attention
        sd[p + "self_attn.in_proj_weight"] = w.astype(np.float32)
        sd[p + "self_attn.in_proj_bias"] = (rng.standard_normal(3 * d) * 0.05).astype(np.float32)
        lin(p + "self_attn.out_proj", d, d)
        sd[p + "multihead_attn.in_proj_weight"] = (rng.standard_normal((3 * d, d)) / np.sqrt(d)).astype(np.float32)
        sd[p + "multihead_attn.in_proj_bias"] = (rng.standard_normal(3 * d) * 0.05).astype(np.float32)
        lin(p + "multihead_attn.out_proj", d, d)
        lin(p + "linear1", ff, d, gain=1.2)
        lin(p + "linear2", d, ff, gain=1.2)
        for n in ("norm1", "norm2", "norm3"):
            sd[p + n + ".weight"] = (1.0 + 0.1 * rng.standard_normal(d)).astype(np.float32)
            sd[p + n + ".bias"] = (0.1 * rng.standard_normal(d)).astype(np.float32)
    if "text" in cfg["cond_mode"]:
        lin("embed_text", d, cfg.get("clip_dim", 512))
    if "action" in cfg["cond_mode"]:
        sd["embed_action.action_embedding"] = rng.standard_normal((cfg["num_actions"], d)).astype(np.float32)
    lin("output_process.poseFinal", F, d, gain=1.0)
    return sd


FAMILIES = ("heavy_tailed", "outlier_channels", "peaky_attention", "big_output", "small_signal", "hostile")


def make_state_dict_family(cfg, family, seed=0):
    """Stress families of synthetic checkpoints for the precision claims (tests: test_stress_family_checkpoints). The i.i.d.-Gaussian
    default above is one point in checkpoint space; trained transformers differ from it in known ways, each of which moves rounding
    error differently: heavy-tailed weights (a few large products dominate a dot product), outlier channels with large LayerNorm gains (the
    'massive activation' channels: an operand's rounding is relative, so a large channel carries a large absolute error into every GEMM that
    reads it), near-one-hot attention (scores far apart: the softmax amplifies score error), a high-gain output projection (multiplies whatever
    error the last layer carries) and a small-signal model (everything near the LayerNorm epsilon's regime). Each is the default checkpoint of
    `seed` with one transformation, so the reference key set and shapes are unchanged. "hostile" stacks them until plain 16-bit operands are
    not enough: the checkpoint the calibration's refusal path is tested with."""
    sd = make_state_dict(cfg, seed=seed)
    if family in (None, "", "gaussian"):
        return sd
    rng = np.random.Generator(np.random.PCG64(7919 + seed))
    d, L = cfg["latent_dim"], cfg["layers"]
    lin_keys = [k for k in sd if k.endswith(".weight") and sd[k].ndim == 2 and "norm" not in k] + [k for k in sd if k.endswith("in_proj_weight")]
    ln_keys = [k for k in sd if ".norm" in k and k.endswith(".weight")]

    def heavy():
        for k in lin_keys:   # Student-t, nu = 3, at the row scale of the Gaussian draw it replaces
            w = sd[k]
            row = w.std(axis=1, keepdims=True)
            sd[k] = (rng.standard_t(3, w.shape) / np.sqrt(3.0) * row).astype(np.float32)

    def outliers(n, gain, compensate):
        # LayerNorm gain AND shift x gain on n channels of every norm: the residual stream carries a few channels `gain` times larger than the rest
        # (they dominate the next LayerNorm's statistics). compensate: the weights that READ those channels - linear1 behind norm2, the next
        # layer's in_proj / the output projection behind norm3 - are 1 / gain there, as training leaves them (the sub-layers then see O(1)
        # inputs). Without it the denoiser's gain is so large that the sampling loop is a chaotic map: see test_parity_presupposes_...
        ch = rng.choice(d, size=n, replace=False)
        for k in ln_keys:
            for kk in (k, k[: -len("weight")] + "bias"):
                g = sd[kk].copy()
                g[ch] *= gain
                sd[kk] = g
        if compensate:
            readers = [f"seqTransDecoder.layers.{l}.linear1.weight" for l in range(L)] + \
                      [f"seqTransDecoder.layers.{l}.self_attn.in_proj_weight" for l in range(1, L)] + ["output_process.poseFinal.weight"]
            for k in readers:
                w = sd[k].copy()
                w[:, ch] /= gain
                sd[k] = w

    def peaky(gain):
        for l in range(L):
            k = f"seqTransDecoder.layers.{l}.self_attn.in_proj_weight"
            w = sd[k].copy()
            w[: 2 * d] *= gain / 2.5
            sd[k] = w

    if family == "heavy_tailed":
        heavy()
    elif family == "outlier_channels":
        outliers(6, 2.0, True)
    elif family == "outlier_channels_uncompensated":
        outliers(6, 8.0, False)
    elif family == "peaky_attention":
        peaky(6.0)
    elif family == "big_output":
        sd["output_process.poseFinal.weight"] = sd["output_process.poseFinal.weight"] * np.float32(3.0)
    elif family == "small_signal":
        for k in lin_keys:
            sd[k] = sd[k] * np.float32(0.3)
    elif family == "hostile":
        heavy()
        outliers(24, 64.0, False)
        peaky(10.0)
        sd["output_process.poseFinal.weight"] = sd["output_process.poseFinal.weight"] * np.float32(8.0)
    else:
        raise ValueError(f"unknown checkpoint family {family!r}: {FAMILIES}")
    return sd


def _random_rot6d(rng, shape):
    """rot6d (first two rows of a rotation matrix, row-major) of uniformly random rotations."""
    q = rng.standard_normal(shape + (4,))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    r0 = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1)
    r1 = np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1)
    return np.concatenate([r0, r1], -1)


def make_cmotion(cfg, batch, seed=1):
    """Actor motion [B, njoints, 6, T] fp32: rot6d joints + last row = translation [tx,ty,tz,0,0,0]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    J, T = cfg["njoints"], cfg["num_frames"]
    assert cfg["nfeats"] == 6
    rot = _random_rot6d(rng, (batch, T, J - 1))                   # [B,T,J-1,6]
    tr = np.zeros((batch, T, 1, 6))
    tr[..., 0, :3] = rng.uniform(-1, 1, (batch, T, 3))
    cm = np.concatenate([rot, tr], axis=2)                          # [B,T,J,6]
    return np.ascontiguousarray(cm.transpose(0, 2, 3, 1)).astype(np.float32)


def make_noise_tape(cfg, batch, steps, seed=10):
    """[steps+1, B, J, 6, T] fp32 N(0,1): entry 0 is x_T, entry k the k-th per-step draw."""
    rng = np.random.Generator(np.random.PCG64(seed))
    shape = (steps + 1, batch, cfg["njoints"], cfg["nfeats"], cfg["num_frames"])
    return rng.standard_normal(shape, dtype=np.float32)


def make_actions(cfg, batch, seed=2):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, max(cfg["num_actions"], 1), (batch, 1)).astype(np.int64)


def make_text_features(cfg, batch, seed=3):
    """Stand-in for CLIP ViT-B/32 text features [B, 512] (CLIP itself is out of scope)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    f = rng.standard_normal((batch, cfg.get("clip_dim", 512)))
    f /= np.linalg.norm(f, axis=-1, keepdims=True)
    return f.astype(np.float32)


def make_stgcn_state_dict(A, num_class=26, in_channels=12, num_person=2, seed=0):
    """Synthetic checkpoint of the ST-GCN evaluator with the reference's state_dict keys
    (eval/a2m/recognition/models/stgcn.py:44-73,183-219; stgcnutils/tgcn.py:51-59). `A` [K, V, V] is the adjacency buffer
    the reference registers (stgcn.py:42); BatchNorm running statistics are non-trivial so that folding them is exercised."""
    rng = np.random.Generator(np.random.PCG64(seed))
    K, V = int(A.shape[0]), int(A.shape[1])
    sd = {"A": np.asarray(A, dtype=np.float32)}

    def bn(prefix, c):
        sd[prefix + ".weight"] = (1.0 + 0.2 * rng.standard_normal(c)).astype(np.float32)
        sd[prefix + ".bias"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        sd[prefix + ".running_mean"] = (0.2 * rng.standard_normal(c)).astype(np.float32)
        sd[prefix + ".running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        sd[prefix + ".num_batches_tracked"] = np.array(100, dtype=np.int64)

    def conv(prefix, co, ci, kt, gain=1.0):
        sd[prefix + ".weight"] = (rng.standard_normal((co, ci, kt, 1)) * (gain / np.sqrt(ci * kt))).astype(np.float32)
        sd[prefix + ".bias"] = (0.05 * rng.standard_normal(co)).astype(np.float32)

    bn("data_bn", in_channels * V)
    chans = [(in_channels // num_person, 64, 1), (64, 64, 1), (64, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 1), (128, 128, 1),
             (128, 256, 2), (256, 256, 1), (256, 256, 1)]
    for i, (ci, co, stride) in enumerate(chans):
        p = f"st_gcn_networks.{i}."
        conv(p + "gcn.conv", co * K, ci, 1, gain=1.5)
        bn(p + "tcn.0", co)
        conv(p + "tcn.2", co, co, 9, gain=1.5)
        bn(p + "tcn.3", co)
        if i > 0 and (ci != co or stride != 1):
            conv(p + "residual.0", co, ci, 1)
            bn(p + "residual.1", co)
        sd[f"edge_importance.{i}"] = (1.0 + 0.3 * rng.standard_normal((K, V, V))).astype(np.float32)
    conv("fcn", num_class, 256, 1)
    return sd


def build_model(cfg, sd, resp="", precision=None, device="cuda:0", noise_schedule="cosine", sigma_small=True, x3_tail=None, engine_options=None, f16_steps=None):
    """(model, diffusion) from regennet_amd for a synth config + checkpoint dict (reference key names): the same
    constructor calls the reference factory makes (utils/model_util.py:66-117), for tests, bench.py and tools.
    x3_tail=None here means the engine's default rule — it was derived on exactly these synthetic checkpoints;
    a model that loads a checkpoint through the plain factory path measures its switch point instead ("auto")."""
    import torch

    from .diffusion import gaussian_diffusion as gd
    from .diffusion.respace import SpacedDiffusion, space_timesteps
    from .model.cmdm import CMDM
    from .utils.model_util import load_model_wo_clip

    kw = {} if precision is None else {"precision": precision}
    model = CMDM("", cfg["njoints"], cfg["nfeats"], cfg["num_actions"], True, "rot6d", True, True,
                 num_frames=cfg["num_frames"], latent_dim=cfg["latent_dim"], ff_size=cfg["ff_size"],
                 num_layers=cfg["layers"], num_heads=cfg["num_heads"], dropout=0.1, activation="gelu",
                 data_rep="rot6d", dataset=cfg["dataset"], arch="online", cm_mode=cfg["cm_mode"], body_model="smplx",
                 cond_mode=cfg["cond_mode"], cond_mask_prob=cfg["cond_mask_prob"], action_emb="tensor",
                 emb_trans_dec=cfg.get("emb_trans_dec", False), wo_pos_emb=cfg.get("wo_pos_emb", False),
                 x3_tail=x3_tail, engine_options=engine_options, f16_steps=f16_steps, **kw)
    load_model_wo_clip(model, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.x3_tail = x3_tail
    model.to(device)
    model.eval()
    diffusion = SpacedDiffusion(use_timesteps=space_timesteps(1000, resp or [1000]),
                                betas=gd.get_named_beta_schedule(noise_schedule, 1000, 1.0),
                                model_mean_type=gd.ModelMeanType.START_X,
                                model_var_type=gd.ModelVarType.FIXED_SMALL if sigma_small else gd.ModelVarType.FIXED_LARGE,
                                loss_type=gd.LossType.MSE, rescale_timesteps=False)
    return model, diffusion
