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


# ---- the multi-GPU sampling entry point (sample/cgenerate.py under torchrun) with the engine stubbed at _lib.Engine -----------
class FakeEngine:
    """Stands in for regennet_amd._lib.Engine on a CPU-only host: same methods, deterministic arithmetic that depends on
    everything the real engine's result depends on (weights blob, condition, seed, GLOBAL sample index, step index)."""
    requires_gpu = False

    def __init__(self, cfg, max_batch, device_index, precision="f32"):
        self.cfg, self.max_batch, self.precision = dict(cfg), int(max_batch), precision
        self.schedule_id, self._w, self.blob, self.calls = None, {}, None, []

    def load_weight(self, key, array):
        self._w[key] = float(np.asarray(array, dtype=np.float64).sum())

    def finalize(self):
        vals = np.array([self._w[k] for k in sorted(self._w)], dtype=np.float32)
        self.blob = torch.from_numpy(vals.view(np.uint8).copy())

    def weight_blob(self):
        self.calls.clear()                       # whoever takes the blob may overwrite it: schedule and condition must follow
        return self.blob, self.blob.numel()

    def set_small_batch_rows(self, n):
        self.knobs = dict(getattr(self, "knobs", {}), small_batch_rows=int(n))

    def set_x3_tail(self, n):
        self.knobs = dict(getattr(self, "knobs", {}), x3_tail=int(n))

    def set_layers_min_b(self, n):
        self.knobs = dict(getattr(self, "knobs", {}), layers_min_b=int(n))

    def set_f16_steps(self, n):
        self.knobs = dict(getattr(self, "knobs", {}), f16_steps=int(n))

    def set_option(self, key, value):
        self.options = dict(getattr(self, "options", {}), **{key: int(value)})

    def set_const_noise(self, on):
        self.const_noise = bool(on)

    def precision_plan(self, B, guided=False):
        from regennet_amd._lib import default_x3_tail
        return 0, default_x3_tail(self.S, self.cfg["layers"], bool(self.cfg.get("emb_trans_dec", False)))

    def set_schedule(self, tmap, tables, sched_id=None):
        self.schedule_id, self.S = sched_id, len(tmap)
        self.calls.append("schedule")

    def set_condition(self, B, cm, action, text, scale, stream):
        self.cond = cm.reshape(B, -1).mean(dim=1).clone()
        self.calls.append("condition")

    def _wsum(self):
        return float(self.blob.view(torch.float32).double().sum()) * 1e-3

    def randn(self, x, B, seed, sample_offset, stream):
        idx = torch.arange(x[0].numel(), dtype=torch.float32).reshape(x[0].shape)
        for b in range(B):
            x[b] = torch.sin(idx * 0.37 + float(seed % 1000) + (sample_offset + b) * 1.7)

    def sample_range(self, sampler, guided, eta, x, noise, seed, sample_offset, first_index, count, x0, use_graph, clip, stream):
        assert "condition" in self.calls and "schedule" in self.calls, "sampling before condition / schedule were (re)bound"
        self.calls.append("range")
        B = x.shape[0]
        for i in range(first_index, first_index - count, -1):
            for b in range(B):
                x[b] = 0.9 * x[b] + 0.1 * (self._wsum() + float(self.cond[b]) + 0.01 * i + 0.001 * ((sample_offset + b) % 97))

    def gaussian_filter1d(self, x, out, rows, T, sigma, stream):
        out.copy_(x)

    def close(self):
        pass


def _cgen_args(out_dir):
    return ["--synthetic", "--unconstrained", "--guidance_param", "1", "--num_samples", "5", "--num_repetitions", "2", "--layers", "1",
            "--timestep_respacing", "ddim5", "--use_ddim", "--motion_length", "40", "--output_dir", out_dir]


def _cgen_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from regennet_amd import _lib
    from regennet_amd.sample import cgenerate
    _lib.Engine = FakeEngine
    path = cgenerate.main(_cgen_args(out_dir))
    assert (path is not None) == (rank == 0)
    dist.destroy_process_group()


def test_cgenerate_entry_point_shards_broadcasts_and_gathers(tmp_path, monkeypatch):
    """`torchrun -m regennet_amd.sample.cgenerate` control flow on 2 gloo ranks: contiguous shards of num_samples, rank 0's
    checkpoint broadcast (rank 1 starts from a different one), per-rank sampling keyed by the global sample index,
    gather + save on rank 0 — and the saved result equals the single-process run (world-size invariance)."""
    from regennet_amd import _lib
    from regennet_amd.sample import cgenerate
    monkeypatch.setattr(_lib, "Engine", FakeEngine)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    one = np.load(cgenerate.main(_cgen_args(str(tmp_path / "w1"))), allow_pickle=True).item()
    mp.spawn(_cgen_worker, args=(2, _free_port(), str(tmp_path / "w2")), nprocs=2, join=True)
    two = np.load(str(tmp_path / "w2" / "results.npy"), allow_pickle=True).item()
    assert one["output"].shape == (10, 56, 6, 40) and two["world_size"] == 2 and one["world_size"] == 1
    assert np.array_equal(one["cmotion"], two["cmotion"])
    assert np.allclose(one["output"], two["output"], atol=1e-6), np.abs(one["output"] - two["output"]).max()
    assert np.abs(one["output"][0] - one["output"][3]).max() > 1e-4      # samples differ (global index, condition)


def _cgen_worker_small(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from regennet_amd import _lib
    from regennet_amd.sample import cgenerate
    _lib.Engine = FakeEngine
    args = [a for a in _cgen_args(out_dir)]
    args[args.index("--num_samples") + 1] = "1"                # one motion for two ranks: rank 1's shard is EMPTY
    path = cgenerate.main(args)
    assert (path is not None) == (rank == 0)
    dist.destroy_process_group()


def test_cgenerate_with_an_empty_shard_does_not_hang_in_the_calibration(tmp_path):
    """num_samples < world size: rank 1 samples nothing and never reaches the sampler's calibration. The switch point of the precision
    schedule (x3_tail="auto", the default for a loaded checkpoint) is agreed at start-up, where every rank passes
    (diffusion.agree_x3_tail: the empty rank contributes 0) - a collective inside the sampling call would leave rank 0 waiting forever
    (mp.spawn would time the test out)."""
    mp.spawn(_cgen_worker_small, args=(2, _free_port(), str(tmp_path / "w2")), nprocs=2, join=True)
    two = np.load(str(tmp_path / "w2" / "results.npy"), allow_pickle=True).item()
    assert two["output"].shape == (2, 56, 6, 40) and two["world_size"] == 2


def _sync_worker(rank, world, port):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from regennet_amd import _lib, synth
    from regennet_amd.model.cfg_sampler import ClassifierFreeSampleModel
    _lib.Engine = FakeEngine
    dev = dist_util.setup_dist()
    cfg = synth.get_config("tiny")
    model, _ = synth.build_model(cfg, synth.make_state_dict(cfg, seed=0 if rank == 0 else 77), precision="bf16_x3tail", device="cpu")
    stale, _d = model._get_engine(2)                             # an engine packed BEFORE the synchronisation (from this rank's own values)
    n_coll = []
    real = dist.broadcast
    dist.broadcast = lambda *a, **k: (n_coll.append(1), real(*a, **k))[1]
    nbytes = dist_util.sync_model_weights(ClassifierFreeSampleModel(model), 0)   # (through the guidance wrapper: the inner module owns the weights)
    ref = synth.make_state_dict(cfg, seed=0)
    assert len(n_coll) == 1 and nbytes == 4 * sum(v.size for k, v in ref.items() if k != "sequence_pos_encoder.pe")   # ONE collective, `pe` once
    for k, v in model.state_dict().items():
        assert torch.equal(v, torch.from_numpy(ref[k])), k       # every rank now holds rank 0's checkpoint, buffers included
    # ... and everything a rank builds afterwards is local: a collective from here on would be a bug (the other rank is not there)
    dist.broadcast = lambda *a, **k: (_ for _ in ()).throw(AssertionError("a collective behind the start-up synchronisation"))
    e1, _d = model._get_engine(2)
    assert e1 is not stale                                       # the pre-synchronisation engine was dropped
    if rank == 1:                                                # one rank alone: a larger batch, another length
        e2, _d = model._get_engine(5)
        e3, _d = model._get_engine(5, T=cfg["num_frames"] - 2)
        assert e2 is not e1 and e3 is not e2 and torch.equal(e3.blob, e1.blob)
    dist.broadcast = real
    blobs = [torch.zeros_like(e1.blob) for _ in range(world)]
    dist.all_gather(blobs, e1.blob)
    assert torch.equal(blobs[0], blobs[1])                       # the packed blobs agree across the ranks without ever having been exchanged
    dist.barrier()
    dist.destroy_process_group()


def test_one_flat_broadcast_synchronises_the_module_and_every_later_engine_is_local():
    """Multi-GPU start-up (utils/dist_util.py:54-83's counterpart): ONE collective - the module's parameters and buffers as a flat fp32 buffer
    (dist_util.sync_model_weights) - after which every rank packs its engines locally; an engine a rank builds later and alone (larger batch,
    another length) issues no collective and still agrees with the other ranks' engines."""
    mp.spawn(_sync_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_a_failed_engine_build_leaves_the_cache_as_it_was(monkeypatch):
    """The engine a rebuild was to replace goes back into the cache when the build throws (out of memory, a refused checkpoint), and the
    half-built engine is closed."""
    from regennet_amd import _lib, synth
    monkeypatch.setattr(_lib, "Engine", FakeEngine)
    cfg = synth.get_config("tiny")
    model, _ = synth.build_model(cfg, synth.make_state_dict(cfg, seed=0), precision="bf16_x3tail", device="cpu")
    e1, _dev = model._get_engine(2)
    closed = []

    class Refusing(FakeEngine):
        def finalize(self):
            raise RuntimeError("refused (test)")

        def close(self):
            closed.append(self)

    monkeypatch.setattr(_lib, "Engine", Refusing)
    with pytest.raises(RuntimeError, match="refused"):
        model._get_engine(9)
    assert len(closed) == 1 and model._engines[cfg["num_frames"]] is e1
    monkeypatch.setattr(_lib, "Engine", FakeEngine)
    e2, _dev = model._get_engine(2)
    assert e2 is e1


def test_model_knobs_reach_the_engine(monkeypatch):
    """CMDM(..., x3_tail=, small_batch_rows=) / the attributes of the same name are handed to the engine on every bind
    (-1 = the engine's default), through the _lib.Engine seam on a CPU-only host."""
    from regennet_amd import _lib, synth
    monkeypatch.setattr(_lib, "Engine", FakeEngine)
    cfg = synth.get_config("tiny")
    model, _ = synth.build_model(cfg, synth.make_state_dict(cfg, seed=0), precision="bf16_x3tail", device="cpu")
    eng, _dev = model._get_engine(2)
    assert eng.knobs == {"x3_tail": -1, "f16_steps": -1, "small_batch_rows": -1, "layers_min_b": -1}
    model.x3_tail, model.f16_steps, model.small_batch_rows, model.layers_min_b = 5, 4, 0, 1
    eng2, _dev = model._get_engine(2)
    assert eng2 is eng and eng.knobs == {"x3_tail": 5, "f16_steps": 4, "small_batch_rows": 0, "layers_min_b": 1}


# ---- bench.py launches its own ranks (the counterpart of the reference's rank bootstrap, utils/dist_util.py:20-42) ---------------
def _run_bench(extra_env, *flags):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--engine-stub", "tests.test_dist_cpu:FakeEngine", "--config", "tiny",
                        "--batch", "2", "--steps", "1", "--warmup", "0", "--respacing", "5", *flags],
                       cwd=root, env=env, capture_output=True, text=True, timeout=240)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stdout + p.stderr


def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no torchrun environment re-launches itself as 2 ranks (gloo here, the engine stubbed at
    the _lib.Engine seam) and reports the line for 2 ranks; rank 0's checkpoint reaches rank 1 through the start-up
    broadcast (dist_util.sync_model_weights)."""
    rc, line, log = _run_bench({}, "--gpus", "2")
    assert rc == 0 and line is not None, log
    assert line["n_gpus"] == 2 and line["rccl_world_size"] == 2 and line["backend"] == "gloo", line
    assert line["devices"] == ["rank0:cpu", "rank1:cpu"] and line["config"]["global_batch"] == 4
    assert line["data"].startswith("STUB ENGINE")                     # never mistaken for a measurement


def test_bench_gpus_2_chi3d_shard_and_serial_engine_build():
    """The per-GPU shard of BASELINE configs[3] (Chi3D, 150 frames, 1024 / 8 = 128 motions per rank) through the same self-launch,
    once with the ranks building their engines in turn (REGENNET_SERIAL_ENGINE_BUILD=1): global batch, world size, per-rank devices."""
    for env in ({}, {"REGENNET_SERIAL_ENGINE_BUILD": "1"}):
        rc, line, log = _run_bench(env, "--gpus", "2", "--config", "chi3d", "--batch", "128")
        assert rc == 0 and line is not None, log
        assert line["n_gpus"] == 2 and line["rccl_world_size"] == 2 and line["config"]["global_batch"] == 256, line
        assert line["config"]["batch_per_gpu"] == 128 and line["devices"] == ["rank0:cpu", "rank1:cpu"], line
        assert "chi3d" in line["config"]["workload"] and line["scaling"] == "weak" and line["engine_build_s"] >= 0, line


def test_bench_global_batch_is_the_strong_scaling_mode():
    """`--global-batch N`: ONE batch of N motions sharded over the ranks (contiguous shards like cgenerate's, ragged when N % ranks != 0), reported
    as "scaling": "strong" with value = N x steps / time - next to the weak mode, so that the first 8-GPU run can yield both curves
    (BASELINE configs[3] / [4] are global batches of 1024 / 2048)."""
    rc, line, log = _run_bench({}, "--gpus", "2", "--global-batch", "5")
    assert rc == 0 and line is not None, log
    assert line["scaling"] == "strong" and line["n_gpus"] == 2 and line["config"]["global_batch"] == 5 and line["config"]["batch_per_gpu"] == "2..3", line
    rc, line1, log = _run_bench({}, "--gpus", "1", "--global-batch", "5")
    assert rc == 0 and line1["scaling"] == "strong" and line1["config"]["global_batch"] == 5, log
    rc, _none, log = _run_bench({}, "--gpus", "2", "--global-batch", "1")
    assert rc != 0 and "nothing to sample" in log


class FailingEngine:
    """Engine stub whose construction fails on rank 0 (test_bench_rank_failure_is_a_nonzero_exit)."""
    requires_gpu = False

    def __init__(self, *a, **k):
        if os.environ.get("RANK", "0") == "0":
            raise RuntimeError("rank 0 cannot build its engine (test)")
        FakeEngine.__init__(self, *a, **k)

    def __getattr__(self, name):
        return getattr(FakeEngine, name).__get__(self)


def test_bench_rank_failure_is_a_nonzero_exit():
    """A rank that dies (here: rank 0 while building its engine) must make `bench.py --gpus 2` exit non-zero with no JSON line: the
    driver launches the scaling bench unattended and reads the exit code."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--engine-stub", "tests.test_dist_cpu:FailingEngine", "--config", "tiny",
                        "--batch", "2", "--steps", "1", "--warmup", "0", "--respacing", "5", "--gpus", "2"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode != 0, p.stdout + p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    del json


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    rc, line, log = _run_bench({"WORLD_SIZE": "1", "RANK": "0"}, "--gpus", "2")
    assert rc != 0 and line is None and "refusing" in log, log
