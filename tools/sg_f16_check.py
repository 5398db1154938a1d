"""The recogniser on single fp16 operand planes (SG_F16, rgn_sg_kernels.hip) next to its default split-bf16 arithmetic: features / logits of both against
the reference's own outputs (tests/golden/stgcn.npz), against each other at the evaluation shape, and the forward time of both. Usage: python tools/sg_f16_check.py [N]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from regennet_amd import synth                      # noqa: E402
from regennet_amd.eval import STGCN                 # noqa: E402

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "stgcn.npz"))
sd = synth.make_stgcn_state_dict(g["A"], num_class=26, seed=0)


def model(f16):
    m = STGCN(in_channels=12, num_class=26, num_person=2, graph_args={"layout": "smplx", "strategy": "spatial"}, device="cuda:0")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    m.to("cuda:0").eval()
    if f16:
        m.engine_options["SG_F16"] = 1
    return m


mx, mf = model(False), model(True)
for tag in ("ntu", "chi3d", "one"):
    x = torch.from_numpy(g[f"x_{tag}"]).cuda()
    ref_f, ref_y = g[f"features_{tag}"], g[f"yhat_{tag}"]
    for name, m in (("x3 ", mx), ("f16", mf)):
        b = m({"output": x})
        f = b["features"].reshape(x.shape[0], -1).cpu().numpy()
        y = b["yhat"].cpu().numpy()
        print(f"[{tag:5s} {name}] max |features - reference| = {np.abs(f - ref_f).max():.2e} (|ref| max {np.abs(ref_f).max():.2f}); logits {np.abs(y - ref_y).max():.2e} "
              f"(|ref| max {np.abs(ref_y).max():.2f}); argmax equal {bool((y.argmax(1) == ref_y.argmax(1)).all())}", flush=True)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(0)
for T in (60, 150):
    x = torch.from_numpy(rng.standard_normal((N, 56, 12, T)).astype(np.float32)).cuda()
    out = {}
    for name, m in (("x3", mx), ("f16", mf)):
        for _ in range(2):
            b = m({"output": x})
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for e0, e1 in ev:
            e0.record()
            b = m({"output": x})
            e1.record()
        torch.cuda.synchronize()
        out[name] = (b["features"].reshape(N, -1).clone(), b["yhat"].clone(), float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev])))
    fx, yx, tx = out["x3"]
    ff, yf, tf = out["f16"]
    print(f"[N={N} T={T}] x3 {tx:.3f} ms | f16 {tf:.3f} ms ({tx / tf:.2f}x) | features f16 vs x3: max abs {float((ff - fx).abs().max()):.2e} of |max| {float(fx.abs().max()):.2f}, "
          f"rms rel {float(((ff - fx).pow(2).mean() / fx.pow(2).mean()).sqrt()):.2e}; logits max abs {float((yf - yx).abs().max()):.2e} of {float(yx.abs().max()):.2f}; "
          f"argmax agree {float((yf.argmax(1) == yx.argmax(1)).float().mean()):.4f}", flush=True)
