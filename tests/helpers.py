"""Shared helpers for the parity tests: rebuild the exact inputs the golden fixtures were made from."""
import ast
import hashlib

import numpy as np

from regennet_amd import synth


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def sd_digest(sd):
    return digest(*[sd[k] for k in sorted(sd)])


def fixture_cfg(g):
    over = ast.literal_eval(str(g["over"]))
    return synth.get_config(str(g["cfg_name"]), **over)


def fixture_opts(g):
    return ast.literal_eval(str(g["opts"])) if "opts" in g else {}


def fixture_inputs(g, loop):
    """Returns (cfg, sd, y_numpy, tape_or_x). y_numpy holds numpy arrays."""
    cfg = fixture_cfg(g)
    sd = synth.make_state_dict(cfg, seed=0)
    assert sd_digest(sd) == str(g["sd_digest"]), "synthetic checkpoint drifted from the golden fixtures"
    B = int(g["B"])
    y = {"cmotion": synth.make_cmotion(cfg, B, seed=1)}
    if "action" in cfg["cond_mode"]:
        y["action"] = synth.make_actions(cfg, B, seed=2)
    if "text" in cfg["cond_mode"]:
        y["text_features"] = synth.make_text_features(cfg, B, seed=3)
    if bool(g["guided"]):
        y["scale"] = np.full((B,), 2.5, dtype=np.float32)
    if loop:
        tape = synth.make_noise_tape(cfg, B, int(g["S"]), seed=10)
        assert digest(tape[0], tape[-1], y["cmotion"]) == str(g["in_digest"])
        return cfg, sd, y, tape
    x = synth.make_noise_tape(cfg, B, 0, seed=11)[0]
    assert digest(x, y["cmotion"]) == str(g["in_digest"])
    return cfg, sd, y, x


def autoreg_inputs(g):
    """Inputs of an auto_regressive fixture (make_golden.gen_autoreg): (cfg, sd, y_numpy, [tape_f for f in range(T)])."""
    cfg = fixture_cfg(g)
    sd = synth.make_state_dict(cfg, seed=0)
    assert sd_digest(sd) == str(g["sd_digest"]), "synthetic checkpoint drifted from the golden fixtures"
    B, S, T = int(g["B"]), int(g["S"]), int(g["T"])
    y = {"cmotion": synth.make_cmotion(cfg, B, seed=1)}
    if "action" in cfg["cond_mode"]:
        y["action"] = synth.make_actions(cfg, B, seed=2)
    tapes = [synth.make_noise_tape(cfg, B, S, seed=100 + f) for f in range(T)]
    assert digest(tapes[0][0], y["cmotion"]) == str(g["in_digest"])
    return cfg, sd, y, tapes


# ---- HIP-side construction (GPU tests) --------------------------------------------------------------
def build_hip(cfg, sd, resp="", precision="f32", device="cuda:0", noise_schedule="cosine", sigma_small=True):
    """(model, diffusion) from regennet_amd for a synth config + synthetic checkpoint."""
    import torch

    from regennet_amd.diffusion import gaussian_diffusion as gd
    from regennet_amd.diffusion.respace import SpacedDiffusion, space_timesteps
    from regennet_amd.model.cmdm import CMDM
    from regennet_amd.utils.model_util import load_model_wo_clip

    model = CMDM("", cfg["njoints"], cfg["nfeats"], cfg["num_actions"], True, "rot6d", True, True,
                 num_frames=cfg["num_frames"], latent_dim=cfg["latent_dim"], ff_size=cfg["ff_size"],
                 num_layers=cfg["layers"], num_heads=cfg["num_heads"], dropout=0.1, activation="gelu",
                 data_rep="rot6d", dataset=cfg["dataset"], arch="online", cm_mode=cfg["cm_mode"], body_model="smplx",
                 cond_mode=cfg["cond_mode"], cond_mask_prob=cfg["cond_mask_prob"], action_emb="tensor",
                 emb_trans_dec=cfg.get("emb_trans_dec", False), wo_pos_emb=cfg.get("wo_pos_emb", False),
                 precision=precision)
    load_model_wo_clip(model, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model.to(device)
    model.eval()
    diffusion = SpacedDiffusion(use_timesteps=space_timesteps(1000, resp or [1000]),
                                betas=gd.get_named_beta_schedule(noise_schedule, 1000, 1.0),
                                model_mean_type=gd.ModelMeanType.START_X,
                                model_var_type=gd.ModelVarType.FIXED_SMALL if sigma_small else gd.ModelVarType.FIXED_LARGE,
                                loss_type=gd.LossType.MSE, rescale_timesteps=False)
    return model, diffusion


def y_to_device(y, device="cuda:0"):
    import torch
    return {k: torch.from_numpy(np.asarray(v)).to(device) for k, v in y.items()}
