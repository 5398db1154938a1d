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
