"""The operand-format axis of the precision schedule: error against the reference's goldens over (fp16 plain steps in front of the tail, split-bf16
tail length), on the forms with fp16 instantiations: the one-kernel decoder stack (k_layers<true>, forced on: LAYERS_MIN_B = 1, small-batch engine
off) for 60 frames, k_qkv_attn_long + k_mlp2 + k_step for 150.      python tools/f16_sweep.py [golden ...]        (GPU box)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import build_hip, fixture_inputs, y_to_device  # noqa: E402

NAMES = sys.argv[1:] or ["ntu_ddpm1000", "ntu_action_ddim100_cfg", "ntu_eval_ddim5", "ntu_eval_5", "ntu_action_eval_ddim5", "ntu_ddpm50", "text150_ddim50_cfg",
                         "chi3d_ddim20_cfg", "chi3d_ddpm20"]
TAILS = (0, 1, 2, 3, 5)
N16 = (0, 2, 4, 8, 16, 10000)
print("rows: fp16 plain steps in front of the tail (10000 = every plain step); columns: split-bf16 tail")
for name in NAMES:
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
    cfg, sd, y, tape = fixture_inputs(g, loop=True)
    guided = bool(g["guided"])
    shape = (int(g["B"]), cfg["njoints"], cfg["nfeats"], cfg["num_frames"])
    for n16 in N16:
        errs = []
        for tail in TAILS:
            model, diffusion = build_hip(cfg, sd, resp=str(g["resp"]), precision="bf16_x3tail/throughput", x3_tail=tail, f16_steps=n16,
                                         engine_options={"LAYERS_MIN_B": 1})
            fm = model
            if guided:
                from regennet_amd.model.cfg_sampler import ClassifierFreeSampleModel
                fm = ClassifierFreeSampleModel(model)
            fn = diffusion.p_sample_loop if str(g["mode"]) == "ddpm" else diffusion.ddim_sample_loop
            out = fn(fm, shape, clip_denoised=False, model_kwargs={"y": y_to_device(y)}, noise_tape=torch.from_numpy(tape))
            errs.append(float(np.abs(out.cpu().numpy() - g["final"]).max()))
            plan = model._engine.precision_plan(shape[0], guided)
            model._engine.close()
        print(f"{name:24s} f16_steps {n16:5d}: " + "  ".join(f"tail {t}: {e:.2e}" for t, e in zip(TAILS, errs)) + f"   (last plan: f16 {plan[0]}, tail {plan[1]})", flush=True)
