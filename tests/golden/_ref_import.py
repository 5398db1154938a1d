"""Import harness for the upstream reference (ONLY usable in the build container).

The reference (a pure-Python/PyTorch project) lives read-only at /root/reference and
can never travel to the GPU box. This module makes its hot-path modules importable
here so that `make_golden.py` can run the *real* reference CPU path and record
golden input/output vectors (data only) under tests/golden/.

Missing third-party packages (clip, timm, smplx) are replaced by inert stand-ins that
are never executed on the sampling path: CLIP is only touched in text mode (where we
patch `encode_text` to return supplied features), DropPath is a dead import, and the
SMPL-X layer is only used by rot2xyz post-processing (out of scope).
"""
import sys
import types

import numpy as np
import torch.nn as nn

REF_ROOT = "/root/reference"


def install():
    if getattr(install, "_done", False):
        return
    if not hasattr(np, "float"):
        np.float = float  # data_loaders/humanml/common/quaternion.py:13 uses the removed alias
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    clip = types.ModuleType("clip")
    clip.load = lambda *a, **k: (nn.Identity(), None)   # frozen stand-in; encode_text is patched
    clip.tokenize = None
    clip.model = types.SimpleNamespace(convert_weights=lambda m: None)
    sys.modules["clip"] = clip

    tl = types.ModuleType("timm.models.layers")
    tl.DropPath = type("DropPath", (nn.Identity,), {})
    sys.modules.update({
        "timm": types.ModuleType("timm"),
        "timm.models": types.ModuleType("timm.models"),
        "timm.models.layers": tl,
    })

    class _Layer(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self.num_betas = 10

    sx = types.ModuleType("smplx")
    sx.SMPLLayer = sx.SMPLXLayer = _Layer
    lbs = types.ModuleType("smplx.lbs")
    lbs.vertices2joints = None
    sys.modules.update({"smplx": sx, "smplx.lbs": lbs})
    install._done = True


def build_reference(cfg, state_dict, timestep_respacing=""):
    """Construct the reference CMDM + SpacedDiffusion for a config dict (see regennet_amd.synth)."""
    install()
    import contextlib
    import io

    import torch
    from diffusion import gaussian_diffusion as gd
    from diffusion.respace import SpacedDiffusion, space_timesteps
    from model.cmdm import CMDM

    with contextlib.redirect_stdout(io.StringIO()):
        model = CMDM(
            "", cfg["njoints"], cfg["nfeats"], cfg["num_actions"], True, "rot6d", True, True,
            num_frames=cfg["num_frames"], latent_dim=cfg["latent_dim"], ff_size=cfg["ff_size"],
            num_layers=cfg["layers"], num_heads=cfg["num_heads"], dropout=0.1, activation="gelu",
            data_rep="rot6d", dataset=cfg["dataset"], arch="online", cm_mode=cfg["cm_mode"],
            body_model="smplx", cond_mode=cfg["cond_mode"], cond_mask_prob=cfg["cond_mask_prob"],
            action_emb="tensor", emb_trans_dec=cfg.get("emb_trans_dec", False),
            wo_pos_emb=cfg.get("wo_pos_emb", False),
        )
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in state_dict.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert len(unexpected) == 0, unexpected
    assert all(k.startswith("clip_model.") for k in missing), missing
    model.eval()

    steps = 1000
    betas = gd.get_named_beta_schedule(cfg.get("noise_schedule", "cosine"), steps, 1.0)
    diffusion = SpacedDiffusion(
        use_timesteps=space_timesteps(steps, timestep_respacing or [steps]),
        betas=betas,
        model_mean_type=gd.ModelMeanType.START_X,
        model_var_type=gd.ModelVarType.FIXED_SMALL if cfg.get("sigma_small", True) else gd.ModelVarType.FIXED_LARGE,
        loss_type=gd.LossType.MSE,
        rescale_timesteps=False,
        data_rep="rot6d",
        num_person=1,
        body_model="smplx",
    )
    return model, diffusion


class NoiseTape:
    """Replace gaussian_diffusion's th.randn / th.randn_like with a recorded tape.

    Draw order in the reference [gaussian_diffusion.py:706,544,785]: one x_T draw, then one
    draw per step (DDPM and DDIM alike, including the final step whose noise is masked out).
    """

    def __init__(self, tape):
        self.tape = tape
        self.pos = 0

    def __enter__(self):
        install()
        import torch
        from diffusion import gaussian_diffusion as gd
        self._gd = gd
        self._orig = gd.th

        tape = self

        class _Th:
            def __getattr__(self, name):
                return getattr(torch, name)

            def randn(self, *shape, **kw):
                return tape._next(shape)

            def randn_like(self, x, **kw):
                return tape._next(tuple(x.shape))

        gd.th = _Th()
        return self

    def _next(self, shape):
        import torch
        out = torch.from_numpy(np.asarray(self.tape[self.pos])).clone()
        assert tuple(out.shape) == tuple(shape), (out.shape, shape)
        self.pos += 1
        return out

    def __exit__(self, *exc):
        self._gd.th = self._orig
        return False
