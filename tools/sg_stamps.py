"""Cycle stamps of k_sg_tconv<256, 256> (a -DRGN_SG_PROF build of the library in place of regennet_amd/libregennet_hip.so; tools/r05_tconv_stamps.sh):
per k-step of workgroup 0, averaged over its 8 waves: vmcnt wait | barrier wait | first MFMA group + DMA issue | rest of the MFMA groups."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regennet_amd import _lib, synth  # noqa: E402
from regennet_amd.eval import STGCN  # noqa: E402

A = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "stgcn.npz"))["A"].astype(np.float32)
model = STGCN(in_channels=12, num_class=26, num_person=2, graph_args={"layout": "smplx"}, device="cuda:0")
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_stgcn_state_dict(A, num_class=26, seed=0).items()}, strict=True)
model = model.to("cuda:0").eval()
x = torch.randn(256, 56, 12, 60, device="cuda:0")
for _ in range(3):
    model({"output": x})
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (64 * 8 * 5))()
lib = _lib.load()
assert lib.rgn_debug_sg_prof(buf) == 0
t = np.array(buf, dtype=np.int64).reshape(64, 8, 5)
d = np.stack([t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2], t[:, :, 4] - t[:, :, 3]], -1).astype(np.float64)   # [step][wave][4]
step = (t[1:, :, 0] - t[:-1, :, 0]).astype(np.float64)
print("k-step (top to top), cycles of the stamp clock: mean %.0f  min %.0f  max %.0f" % (step[8:].mean(), step[8:].min(), step[8:].max()))
print("per step, mean over waves: vmcnt wait | barrier | group 0 + DMA issue | groups 1.. ")
for s in range(8, 40):
    print("  step %2d (tap %d): " % (s, s % 9) + "  ".join("%6.0f" % v for v in d[s].mean(0)) + "   | per wave vmcnt: " + " ".join("%5.0f" % v for v in d[s, :, 0]))
print("means over steps 8..63:", d[8:].mean((0, 1)).round(0))
