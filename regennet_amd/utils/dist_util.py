"""Distributed helpers — MI355X counterpart of the reference's `utils/dist_util.py`.

The reference rendezvouses over mpi4py and pins 4 GPUs per node (dist_util.py:15-42). Here one process
drives one GPU (8 per node), rendezvous comes from torchrun's environment, and the backend is
`nccl` (= RCCL on ROCm, xGMI inside a node) or `gloo` on CPU-only hosts. Sampling shards only the
batch axis: the single start-up collective is ONE broadcast of the module's parameters and buffers as a
flat fp32 buffer (`sync_model_weights`, replacing the per-tensor `sync_params`, dist_util.py:77-83) plus an
optional all_gather of the outputs; every rank then packs its engines from the synchronised module locally.
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


def sync_model_weights(model, src=0):
    """THE start-up collective of a multi-GPU run: rank `src`'s checkpoint - every parameter and buffer of the module, flattened into one fp32
    buffer on the device (115 MB for the shipped 8-layer model) - is broadcast over RCCL / xGMI and written back into every rank's module
    (replaces the per-tensor `sync_params` of the reference, utils/dist_util.py:77-83; only rank `src` needs to have read the checkpoint file).
    Every engine a rank builds afterwards - at start-up, or later and alone for another sequence length or a larger batch - packs its own blob
    from the synchronised module (0.4 s of one host core), so nothing downstream of this call is a collective and no rank ever depends on another
    rank having built the same engine. Rounds 1-5 broadcast the PACKED blob of one engine instead (fp32 + three 16-bit layouts: 204 MB, now 237)
    and had to hand it from engine to engine on every rebuild. Returns the bytes broadcast (0 in a single process)."""
    if not collectives_active():
        return 0
    inner = getattr(model, "model", model)                     # (a ClassifierFreeSampleModel wraps the module that owns the weights)
    seen, tensors = set(), []
    for t in inner.state_dict(keep_vars=True).values():       # parameters and buffers, deterministic order; shared storage (the two `pe` keys) once
        if t.data_ptr() not in seen:
            seen.add(t.data_ptr())
            tensors.append(t)
    flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors])
    dist.broadcast(flat, src)
    off = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
    inner._engine_stale = True                                  # engines packed from the old values are obsolete
    return flat.numel() * 4


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
