"""How much of the fp16 operand phase's error is WEIGHT rounding (a fixed perturbation of the model, coherent from step to step) and how much ACTIVATION rounding
(fresh every step)? The same sampling runs on a checkpoint whose 2-D weights are already fp16 numbers (the engine's fp16 packing is then exact for every weight it
does not fold) against the oracle ON THAT CHECKPOINT, next to the unrounded checkpoint against its own oracle. If activation rounding alone measures like the
split-bf16 tail, a two-MFMA fp16 form (a_h w_h + a_h w_l: exact weights, rounded activations) could replace the three-MFMA tail.   python tools/experiments/f16_weight_split.py   (GPU box)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import regennet_oracle as orc      # noqa: E402  (tools: the checker)
from regennet_amd import synth                 # noqa: E402
from tests.helpers import build_hip            # noqa: E402

torch.set_num_threads(32)


def fp16_weights(sd):
    out = {}
    for k, v in sd.items():
        a = np.asarray(v)
        out[k] = a.astype(np.float16).astype(np.float32) if (a.dtype == np.float32 and a.ndim >= 2) else a
    return out


def run(cfg, sd, resp, mode, B, seed, **kw):
    S = len(orc.make_schedule("cosine", resp)[0])
    tape = synth.make_noise_tape(cfg, B, S, seed=seed)
    y = {"cmotion": synth.make_cmotion(cfg, B, seed=seed + 1)}
    if cfg.get("cond_mode") == "action":
        y["action"] = synth.make_actions(cfg, B, seed=seed + 2)
    ref = orc.sample_loop(sd, cfg, orc.make_schedule("cosine", resp), tape, {k: torch.from_numpy(v) for k, v in y.items()}, mode=mode).numpy()
    res = {}
    for name, (n16, tail) in {"default 8+2": (8, 2), "fp16 to the end": (10000, 0), "fp16 + 1 split": (10000, 1), "split-bf16 throughout": (0, 10000)}.items():
        model, diffusion = build_hip(cfg, sd, resp=resp, precision="bf16_x3tail/throughput" if tail != 10000 else "bf16x3/throughput", x3_tail=None if tail == 10000 else tail,
                                     f16_steps=n16, engine_options={"LAYERS_MIN_B": 1})
        fn = diffusion.p_sample_loop if mode == "ddpm" else diffusion.ddim_sample_loop
        out = fn(model, (B, cfg["njoints"], cfg["nfeats"], cfg["num_frames"]), clip_denoised=False, model_kwargs={"y": {k: torch.from_numpy(v).cuda() for k, v in y.items()}},
                 noise_tape=torch.from_numpy(tape))
        res[name] = float(np.abs(out.cpu().numpy() - ref).max())
        model._engine.close()
    return res


for cname, resp, mode, B in (("ntu", "ddim5", "ddpm", 8), ("ntu", "50", "ddpm", 4), ("ntu_action", "ddim5", "ddpm", 8), ("ntu", "ddim20", "ddim", 4)):
    cfg = synth.get_config(cname)
    for fam in (None, "heavy_tailed", "big_output"):
        sd = synth.make_state_dict(cfg, seed=0) if fam is None else synth.make_state_dict_family(cfg, fam, seed=0)
        a = run(cfg, sd, resp, mode, B, 100)
        b = run(cfg, fp16_weights(sd), resp, mode, B, 100)
        print(f"{cname:10s} {resp:6s} {mode} B={B} {fam or 'gaussian':13s} | checkpoint as is: " + ", ".join(f"{k} {v:.2e}" for k, v in a.items()), flush=True)
        print(f"{'':46s} | weights = fp16 numbers: " + ", ".join(f"{k} {v:.2e}" for k, v in b.items()), flush=True)
