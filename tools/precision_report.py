"""GPU tool: max-abs error of each precision mode vs the golden reference outputs (prints a small table)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import build_hip, fixture_inputs, y_to_device  # noqa: E402

for name in ["ntu_fwd", "ntu_ddpm50", "ntu_action_ddim100_cfg", "text150_ddim50_cfg", "ntu_ddpm1000"]:
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    loop = "final" in g
    cfg, sd, y, xin = fixture_inputs(g, loop=loop)
    row = [name]
    for prec in ["f32", "bf16x3", "bf16"]:
        model, diffusion = build_hip(cfg, sd, resp=str(g["resp"]) if loop else "", precision=prec)
        fm = model
        if bool(g["guided"]):
            from regennet_amd.model.cfg_sampler import ClassifierFreeSampleModel
            fm = ClassifierFreeSampleModel(model)
        yd = y_to_device(y)
        if loop:
            fn = diffusion.p_sample_loop if str(g["mode"]) == "ddpm" else diffusion.ddim_sample_loop
            out = fn(fm, (int(g["B"]), cfg["njoints"], cfg["nfeats"], cfg["num_frames"]), clip_denoised=False,
                     model_kwargs={"y": yd}, noise_tape=torch.from_numpy(xin)).cpu().numpy()
            err = np.abs(out - g["final"]).max()
        else:
            err = 0.0
            for i, t in enumerate(g["ts"]):
                o = fm(torch.from_numpy(xin).cuda(), torch.full((xin.shape[0],), int(t), dtype=torch.long, device="cuda"), y=yd)
                err = max(err, float(np.abs(o.cpu().numpy() - g["out"][i]).max()))
        row.append(f"{prec}={err:.2e}")
        model._engine.close()
    print("  ".join(row), flush=True)
