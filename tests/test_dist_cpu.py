"""CPU, world_size=2, gloo: the N>1 path of the sampler — batch sharding, the single flat
weight broadcast and the output all_gather. One process per (would-be) GPU, rendezvous on 127.0.0.1."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from regennet_amd.utils import dist_util


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dev = dist_util.setup_dist()
    assert dev.type == "cpu" and dist.is_initialized() and dist.get_backend() == "gloo"
    assert dist_util.world() == (rank, world)
    # 1) one flat broadcast of the "weight blob"
    blob = torch.arange(1000, dtype=torch.uint8) if rank == 0 else torch.zeros(1000, dtype=torch.uint8)
    dist_util.broadcast_flat(blob, 0)
    assert torch.equal(blob, torch.arange(1000, dtype=torch.uint8))
    # 2) sync_params as a single flattened collective
    g = torch.Generator().manual_seed(rank)
    params = [torch.randn(3, 4, generator=g), torch.randn(7, generator=g)]
    dist_util.sync_params(params)
    g0 = torch.Generator().manual_seed(0)
    assert torch.equal(params[0], torch.randn(3, 4, generator=g0)) and torch.equal(params[1], torch.randn(7, generator=g0))
    # 3) ragged batch shards -> all_gather reassembles the global batch in order
    lo, hi = dist_util.shard_bounds(total)
    full = torch.arange(total * 6, dtype=torch.float32).reshape(total, 2, 3)
    got = dist_util.all_gather_samples(full[lo:hi].clone(), total)
    assert torch.equal(got, full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [5, 8])
def test_two_rank_gloo_sharding_and_collectives(total):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, total), nprocs=2, join=True)


def test_shard_bounds_partition_is_contiguous_and_complete():
    for total in (1, 7, 256, 1024, 2048):
        for w in (1, 2, 4, 8):
            b = [dist_util.shard_bounds(total, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_helpers_are_noops():
    t = torch.ones(4)
    assert dist_util.broadcast_flat(t) is t
    assert dist_util.all_gather_samples(t, 4) is t
    dist_util.sync_params([t])
    assert dist_util.shard_bounds(10, 0, 1) == (0, 10)
