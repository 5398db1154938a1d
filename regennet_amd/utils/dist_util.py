"""Distributed helpers — MI355X counterpart of the reference's `utils/dist_util.py`.

The reference rendezvouses over mpi4py and pins 4 GPUs per node (dist_util.py:15-42). Here one process
drives one GPU (8 per node), rendezvous comes from torchrun's environment, and the backend is
`nccl` (= RCCL on ROCm, xGMI inside a node) or `gloo` on CPU-only hosts. Sampling shards only the
batch axis: the single collective is ONE broadcast of the flat packed-weight buffer (replacing the
per-tensor `sync_params`, dist_util.py:77-83) plus an optional all_gather of the outputs.
"""
import os

import torch
import torch.distributed as dist

_device = None


def collectives_active():
    """True when this process takes the multi-rank code paths: a process group of more than one rank - or of ONE rank with
    REGENNET_FORCE_DIST=1, which is how the collectives (RCCL init, blob broadcast, barriers, gathers) are exercised on a 1-GPU box."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or bool(os.environ.get("REGENNET_FORCE_DIST")))


def setup_dist(device=None):
    """setup_dist() (dist_util.py:20-42): bind this process to its GPU; join the process group when launched
    under torchrun (RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT/LOCAL_RANK)."""
    global _device
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        idx = local_rank % torch.cuda.device_count() if device is None else int(device)
        torch.cuda.set_device(idx)
        _device = torch.device(f"cuda:{idx}")
    else:
        _device = torch.device("cpu")
    if (world_size > 1 or os.environ.get("REGENNET_FORCE_DIST")) and not dist.is_initialized():
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if _device.type == "cuda":
            kw["device_id"] = _device
        dist.init_process_group(backend="nccl" if _device.type == "cuda" else "gloo", **kw)
    return _device


def dev():
    """dev() (dist_util.py:45-51)."""
    if _device is not None:
        return _device
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


def shard_bounds(total, rank=None, world_size=None):
    """Contiguous batch shard [lo, hi) of `total` samples for this rank."""
    r, w = world() if rank is None else (rank, world_size)
    base, extra = divmod(total, w)
    lo = r * base + min(r, extra)
    return lo, lo + base + (1 if r < extra else 0)


def device_view(ptr, nbytes, device):
    """Zero-copy uint8 tensor over raw device memory (e.g. the engine's packed weight blob, rgn_weight_blob)."""

    class _Raw:
        __cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

    return torch.as_tensor(_Raw(), device=device)


def broadcast_flat(buf, src=0):
    """Broadcast one flat tensor from `src` (the packed weight blob): one collective over xGMI."""
    if collectives_active():
        dist.broadcast(buf, src)
    return buf


def stream_handle(device):
    """hipStream_t of torch's current stream on `device` as an int (0 on a CPU-only host: gloo tests with a stub engine)."""
    device = torch.device(device)
    return torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0


def synchronize(device=None):
    if torch.cuda.is_available():
        torch.cuda.synchronize(device)


def broadcast_engine_weights(engine, device, src=0):
    """Start-up collective of a multi-GPU run: rank `src`'s packed weight blob (rgn_weight_blob: ONE flat device buffer)
    is broadcast into every rank's blob over RCCL / xGMI — replaces the per-tensor `sync_params` of the reference
    (utils/dist_util.py:77-83). Callers must re-derive everything computed FROM the weights afterwards (the per-schedule
    tables: `engine.schedule_id = None`; the hoisted condition is rebound by every sampling call anyway)."""
    if not collectives_active():
        return
    ptr, nbytes = engine.weight_blob()
    view = ptr if torch.is_tensor(ptr) else device_view(ptr, nbytes, device)   # (a stub engine hands over a tensor)
    synchronize(device)
    dist.broadcast(view, src)
    synchronize(device)
    engine.schedule_id = None


def copy_engine_weights(src_engine, dst_engine, device):
    """Device-to-device copy of one engine's packed weight blob into another engine of the SAME model (same blob layout): what a
    rank does for engines it builds alone after the start-up broadcast (no collective). False if the layouts differ."""
    sp, sn = src_engine.weight_blob()
    dp, dn = dst_engine.weight_blob()
    if int(sn) != int(dn):
        return False
    sv = sp if torch.is_tensor(sp) else device_view(sp, sn, device)
    dv = dp if torch.is_tensor(dp) else device_view(dp, dn, device)
    synchronize(device)
    dv.copy_(sv)
    synchronize(device)
    dst_engine.schedule_id = None
    return True


def sync_params(params):
    """dist_util.py:77-83, but as a single flattened broadcast instead of one per tensor."""
    params = list(params)
    if not collectives_active() or not params:
        return
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    dist.broadcast(flat, 0)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))
            off += n


def all_gather_samples(local, total):
    """Gather per-rank sample shards [b_r, ...] back into [total, ...] on every rank."""
    if not collectives_active():
        return local
    w = dist.get_world_size()
    sizes = [shard_bounds(total, r, w) for r in range(w)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(outs, pad)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(outs, sizes)], dim=0)
