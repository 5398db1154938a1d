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
def build_hip(cfg, sd, resp="", precision="f32", device="cuda:0", noise_schedule="cosine", sigma_small=True, x3_tail=None, engine_options=None, f16_steps=None):
    """(model, diffusion) from regennet_amd for a synth config + synthetic checkpoint.
    precision "<mode>/throughput": the same mode with the small-batch engine (rgn_sb.hip) switched off, so a test-sized
    batch runs the kernels a full-size batch gets."""
    mode, _, engine = precision.partition("/")
    model, diffusion = synth.build_model(cfg, sd, resp=resp, precision=mode, device=device, noise_schedule=noise_schedule,
                                         sigma_small=sigma_small, x3_tail=x3_tail, engine_options=engine_options, f16_steps=f16_steps)
    if engine == "throughput":
        model.small_batch_rows = 0
    return model, diffusion


def y_to_device(y, device="cuda:0"):
    import torch
    return {k: torch.from_numpy(np.asarray(v)).to(device) for k, v in y.items()}
