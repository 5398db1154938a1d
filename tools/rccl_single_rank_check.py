"""One-rank RCCL check on a 1-GPU box (tools only): does torch.distributed's nccl (= RCCL) backend initialise in this image with the environment
bench.py builds for its ranks, and does the collective of the multi-GPU path - dist_util.sync_model_weights: ONE broadcast of the module's
parameters as a flat fp32 buffer - run? (Two ranks cannot share one GPU under RCCL, so N > 1 itself stays unmeasured here.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from regennet_amd import synth  # noqa: E402
from regennet_amd.utils import dist_util  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda:0")
t0 = time.time()
dist.init_process_group("nccl", device_id=dev)
x = torch.ones(1 << 20, device=dev)
dist.broadcast(x, 0)
dist.all_reduce(x)
torch.cuda.synchronize()
print(f"[rccl] backend {dist.get_backend()} world {dist.get_world_size()}: init + broadcast + all_reduce in {time.time() - t0:.2f} s, x[0] = {x[0].item()}")
cfg = synth.get_config("ntu")
model, diffusion = synth.build_model(cfg, synth.make_state_dict(cfg, seed=0), resp="12", precision="bf16_x3tail", device="cuda:0")
sd_before = {k: v.clone() for k, v in model.state_dict().items()}
os.environ["REGENNET_FORCE_DIST"] = "1"                 # (dist_util.collectives_active: a one-rank group takes the multi-rank paths)
t0 = time.time()
nbytes = dist_util.sync_model_weights(model, 0)          # THE start-up collective: the module's parameters + buffers as one flat fp32 buffer
torch.cuda.synchronize()
same = all(torch.equal(v, sd_before[k]) for k, v in model.state_dict().items())
print(f"[rccl] checkpoint: {nbytes / 1e6:.1f} MB of fp32 parameters broadcast as one flat buffer in {1e3 * (time.time() - t0):.1f} ms, values unchanged: {same}")
eng, _ = model._get_engine(4)                            # packed locally from the synchronised module
tt = torch.tensor([3], device=dev, dtype=torch.int64)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)          # the x3_tail="auto" agreement
print(f"[rccl] MAX all_reduce of the calibrated tail: {int(tt.item())}")
dist.barrier()
dist.destroy_process_group()
print("[rccl] ok")
