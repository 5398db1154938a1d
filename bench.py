#!/usr/bin/env python
"""bench.py — sampled motions/sec of the ReGenNet diffusion-sampling hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one complete sampling call (x_T -> x_0) for one batch of synthetic motions: BASELINE.json
configs[1] = NTU120-AS shape [B=256, 56, 6, 60], online/concat/no_cond, 8 layers, 1000-step DDPM per GPU
(weak scaling: every rank samples its own 256 motions; the only collective is the start-up RCCL broadcast
of the packed weight blob). Inputs (condition, weights) are resident in HBM before the timed region; noise
comes from the on-device Philox stream. Prints ONE JSON line on rank 0.

Extra objects: "roofline" for the dominant kernel class (the MFMA GEMMs; durations measured with HIP
events on the engine's stream in a short profiled pass inside this script) and "cpu_baseline" (the oracle
= CPU port of the reference path, timed on this host's cores on a bounded sample, rank 0 / N=1 only).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# minimal algorithmic work of ONE denoiser evaluation for ONE sample (FLOP = 2*MAC)
ALGO_GFLOP_PER_EVAL = {("ntu", "concat"): 2.154, ("ntu", "add"): 2.123, ("chi3d", "concat"): 5.593, ("chi3d", "add"): 5.514}
PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0, "bf16_x3tail": 2500.0, "f16": 2500.0}   # MI355X_MICROARCH.md dense MFMA peaks


def row_check(cfg, sd, a, dev, y, out, seed, lo, plan, fn_name, rows):
    """Motion b of the last timed call against a B = 1 run of the SAME kernel form (model.layers_min_b = 1 where the batch ran the
    one-kernel decoder stack, throughput kernels instead of the small-batch engine) with the motion's global Philox key: the launch the
    headline number rests on - 256 workgroups x 995 steps - is checked at its real shape on every bench run, not only for isfinite."""
    from regennet_amd import synth
    from regennet_amd.model.cfg_sampler import ClassifierFreeSampleModel
    model1, diffusion1 = synth.build_model(cfg, sd, resp=a.respacing, precision=a.precision, device=str(dev), x3_tail=a.x3_tail, f16_steps=a.f16_steps)
    if "sb_gemm" not in plan:                       # (a batch the small-batch engine ran is bit-exact under batch composition by itself)
        model1.small_batch_rows = 0
    if any(k in plan for k in ("layers", "steps_fused")):
        model1.layers_min_b = 1
        if a.guided:      # ... and the batch's guided form: a motion per workgroup (k_layers<true, true>) or an evaluation per workgroup and step
            model1.layers_guided = 2 if "steps_fused" in plan else 0
    fm1 = ClassifierFreeSampleModel(model1) if a.guided else model1
    fn1 = getattr(diffusion1, fn_name)
    worst = 0.0
    for b in rows:
        yb = {k: v[b:b + 1].contiguous() for k, v in y.items()}
        one = fn1(fm1, (1,) + tuple(out.shape[1:]), clip_denoised=False, model_kwargs={"y": yb}, seed=seed, sample_offset=lo + b,
                  use_graph=not a.no_graph)
        worst = max(worst, float((out[b:b + 1] - one).abs().max()))
    model1._engine.close()
    return worst


def _recogniser_graph(K, V):
    """The reference recogniser's own normalised adjacency (3 x 56 x 56, 166 nonzeros; longest lists 1 / 6 / 1 per partition): the `A` buffer of the
    reference model as committed with the golden vectors (tests/golden/stgcn.npz, written by tests/golden/make_golden.py)."""
    A = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "stgcn.npz"))["A"].astype(np.float32)
    assert A.shape == (K, V, V), A.shape
    return A


def bench_stgcn(a):
    """`--config stgcn`: the evaluation harness's recogniser (rgn_stgcn_forward; eval/a2m/recognition/models/stgcn.py:76-123) on N = --batch
    two-person motions of 60 and 150 frames, timed with the same barrier / synchronize bracket, HIP events around every forward, against the
    dense bf16 MFMA peak (its GEMMs are split-bf16: three MFMAs per product, so the ceiling for ALGORITHMIC FLOPs is a third of it). FLOPs:
    the ten st_gcn blocks of stgcn.py:51-62 (graph aggregation on the input channels - counted over the skeleton's nonzeros, as the kernels walk them -
    1x1 convolution over K C_in, 9x1 temporal convolution, residual 1x1 where the block has one), both persons."""
    from regennet_amd import synth
    from regennet_amd.eval import STGCN
    dev = torch.device("cuda:0")
    V, K, M, C = 56, 3, 2, 6
    rng = np.random.default_rng(0)
    A = _recogniser_graph(K, V)
    sd = synth.make_stgcn_state_dict(A, num_class=26, seed=0)
    model = STGCN(in_channels=M * C, num_class=26, num_person=M, graph_args={"layout": "smplx", "strategy": "spatial"}, device=str(dev))
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.to(dev).eval()
    ref_model = None
    if a.recogniser_f16:
        # the fp16 form next to the default arithmetic: same checkpoint, same input, feature / logit differences reported with the line
        ref_model = STGCN(in_channels=M * C, num_class=26, num_person=M, graph_args={"layout": "smplx", "strategy": "spatial"}, device=str(dev))
        ref_model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        ref_model.to(dev).eval()
        model.engine_options["SG_F16"] = 1
    blocks = [(C, 64, 1, False), (64, 64, 1, False), (64, 64, 1, False), (64, 64, 1, False), (64, 128, 2, True), (128, 128, 1, False),
              (128, 128, 1, False), (128, 256, 2, True), (256, 256, 1, False), (256, 256, 1, False)]
    lines = []
    for T in (60, 150):
        N = a.batch
        x = torch.from_numpy(rng.standard_normal((N, V, M * C, T)).astype(np.float32)).to(dev)
        # graph aggregation: the reference contracts densely over (k, v, w) (tgcn.py: einsum 'nkctv,kvw->nctw'); the kernels walk the nonzeros of
        # the 3 x 56 x 56 adjacency (166 here). `flops` - what the roofline uses - counts the EXECUTED aggregation MACs; the dense count is reported next to it
        nnz = int(np.count_nonzero(A))
        mac, mac_dense, t = 0.0, 0.0, T
        for ci, co, st, res in blocks:
            to = (t + st - 1) // st
            rest = t * V * K * ci * co + to * V * 9 * co * co + (to * V * ci * co if res else 0)
            mac += nnz * ci * t + rest
            mac_dense += K * ci * t * V * V + rest
            t = to
        flops, flops_dense = 2.0 * mac * M * N, 2.0 * mac_dense * M * N
        for _ in range(max(a.warmup, 1)):
            model({"output": x})
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
        t0 = time.perf_counter()
        for e0, e1 in evs:
            e0.record()
            model({"output": x})
            e1.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in evs]))
        tf = flops / (ms * 1e-3) / 1e12
        vs_x3 = None
        if ref_model is not None:
            o16, o3 = model({"output": x}), ref_model({"output": x})
            f16_, f3 = o16["features"].reshape(N, -1), o3["features"].reshape(N, -1)
            vs_x3 = {"features_max_abs": float((f16_ - f3).abs().max()), "features_abs_max_of_default": float(f3.abs().max()),
                     "features_rms_rel": float(((f16_ - f3).pow(2).mean() / f3.pow(2).mean()).sqrt()), "logits_max_abs": float((o16["yhat"] - o3["yhat"]).abs().max()),
                     "argmax_agree": float((o16["yhat"].argmax(1) == o3["yhat"].argmax(1)).float().mean())}
        lines.append({"T": T, "N": N, "ms_per_forward": round(ms, 3), "wall_ms_per_forward": round(1e3 * dt / a.steps, 3),
                      "motions_per_s": round(N / (ms * 1e-3), 1), "algo_gflop_per_forward": round(flops / 1e9, 2),
                      "gflop_per_forward_dense_aggregation": round(flops_dense / 1e9, 2), **({"fp16_form_vs_default_arithmetic": vs_x3} if vs_x3 else {}),
                      "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_TFLOPS["bf16x3"], "unit": "TFLOP/s", "frac": round(tf / PEAK_TFLOPS["bf16x3"], 4),
                                   "traffic": None, "note": ("SG_F16: ONE fp16 MFMA per product in blocks 1-9 (ceiling for algorithmic FLOPs = peak); otherwise as the default's note: " if a.recogniser_f16 else "") +
                                                            "FLOPs of one forward (all its launches; graph aggregation counted over the adjacency's nonzeros, as executed - the dense reference count "
                                                            "is gflop_per_forward_dense_aggregation) / its duration; split-bf16 GEMMs: three MFMAs per product, "
                                                            "ceiling for algorithmic FLOPs = peak / 3; the 4 shared zero pad frames per sequence are computed too (not counted)"}})
    print(json.dumps({"metric": "ST-GCN recogniser forward (evaluation harness, SURVEY 8f next-4)", "value": lines[0]["motions_per_s"], "unit": "motions/s",
                      "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": lines[0]["ms_per_forward"], "higher_is_better": True,
                      "scaling": "weak", "vs_baseline": None, "dtype": "fp16 MFMA (block 0's graph convolution split-bf16), fp32 accumulate" if a.recogniser_f16 else "split-bf16 (x3) MFMA, fp32 accumulate",
                      "data": "synthetic", "config": {"workload": f"stgcn: N={a.batch} x [56, 12, 60 | 150]" + (" (SG_F16)" if a.recogniser_f16 else "")},
                      "roofline": lines[0]["roofline"], "per_length": lines}), flush=True)


def bench_eval_pipeline(a):
    """`--config eval_pipeline`: one evaluation batch of eval/eval_cmdm.py's hot loop on the device (eval/a2m/stgcn_eval.py:61-81 + evaluate.py:55-124): sample
    B action-conditioned NTU motions with the reference's shipped schedule (`ddim5` through p_sample_loop), concatenate with the actor motion, run the
    recogniser, keep the features; FID / accuracy statistics once at the end. A step = one batch; the stages are bracketed with HIP events."""
    from regennet_amd import synth
    from regennet_amd.eval import STGCN
    from regennet_amd.eval.fid import calculate_activation_statistics, calculate_fid
    dev = torch.device("cuda:0")
    cfg = synth.get_config("ntu_action")
    model, diffusion = synth.build_model(cfg, synth.make_state_dict(cfg, seed=0), resp="ddim5", precision=a.precision, device=str(dev))
    V, K = 56, 3
    A = _recogniser_graph(K, V)
    rec = STGCN(in_channels=12, num_class=26, num_person=2, graph_args={"layout": "smplx", "strategy": "spatial"}, device=str(dev))
    rec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_stgcn_state_dict(A, num_class=26, seed=0).items()}, strict=True)
    rec.to(dev).eval()
    if a.recogniser_f16:
        rec.engine_options["SG_F16"] = 1
    B = a.batch
    y = {"cmotion": torch.from_numpy(synth.make_cmotion(cfg, B, seed=1)).to(dev), "action": torch.from_numpy(synth.make_actions(cfg, B, seed=2)).to(dev)}
    shape = (B, 56, 6, 60)

    def batch(seed, ev=None):
        if ev:
            ev[0].record()
        sample = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": y}, seed=seed)
        if ev:
            ev[1].record()
        out = rec({"output": torch.cat((y["cmotion"], sample), dim=2)})     # stgcn_eval.py:71
        if ev:
            ev[2].record()
        return out["features"].reshape(B, 256), out["yhat"]

    # ... and with the actor's half of the recogniser kept: every repetition / seed of the reference's evaluation re-samples the reactor for the SAME actor
    # clips, and in eval mode the two persons never meet before the final mean (STGCN.person_features / features_from_persons)
    actor_feats = rec.person_features(y["cmotion"], person=0)

    def batch_cached(seed, ev):
        ev[0].record()
        sample = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": y}, seed=seed)
        ev[1].record()
        out = rec.features_from_persons([actor_feats, sample])
        ev[2].record()
        return out["features"].reshape(B, 256)

    for w in range(max(a.warmup, 1)):
        batch(10 + w)
        batch_cached(10 + w, [torch.cuda.Event(enable_timing=True) for _ in range(3)])
    torch.cuda.synchronize()
    evc = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.steps)]
    tc0 = time.perf_counter()
    fc = [batch_cached(100 + k, evc[k]) for k in range(a.steps)]
    torch.cuda.synchronize()
    dtc = time.perf_counter() - tc0
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.steps)]
    feats = []
    t0 = time.perf_counter()
    for k in range(a.steps):
        feats.append(batch(100 + k, evs[k])[0])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    stats = calculate_activation_statistics(torch.cat(feats))
    fid_self = float(calculate_fid(stats, stats))
    torch.cuda.synchronize()
    t_fid = time.perf_counter() - t1                          # first call of the process: loads the fp64 solver library
    t2 = time.perf_counter()
    stats = calculate_activation_statistics(torch.cat(feats))
    fid_self = float(calculate_fid(stats, stats))
    torch.cuda.synchronize()
    t_fid2 = time.perf_counter() - t2
    ms_s = float(np.mean([e[0].elapsed_time(e[1]) for e in evs])), float(np.mean([e[1].elapsed_time(e[2]) for e in evs]))
    print(json.dumps({"metric": "evaluation batches of eval_cmdm's hot loop (sample ddim5 + ST-GCN features)", "value": round(a.steps * B / dt, 1), "unit": "motions/s", "n_gpus": 1,
                      "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": "sampler: bf16 / fp16 / split-bf16 schedule; recogniser: " + ("fp16 (SG_F16)" if a.recogniser_f16 else "split-bf16"), "data": "synthetic",
                      "config": {"workload": f"ntu_action B={B}: p_sample_loop(ddim5) -> cat(cmotion, sample) -> STGCN features; FID statistics over {a.steps * B} motions once"},
                      "actor_features_cached": {"ms_per_step": round(1e3 * dtc / a.steps, 3), "motions_per_s": round(a.steps * B / dtc, 1),
                                                "recogniser_ms": round(float(np.mean([e[1].elapsed_time(e[2]) for e in evc])), 3),
                                                "max_abs_dev_of_features_vs_two_person_forward": float((fc[-1] - feats[-1]).abs().max()) if feats else None,
                                                "note": "the recogniser evaluates the sampled reactor only; the actor's pooled features were computed once (same seeds: same samples)"},
                      "stage_ms": {"sample_ddim5": round(ms_s[0], 3), "recogniser": round(ms_s[1], 3), "fid_statistics_first_call": round(1e3 * t_fid, 2), "fid_statistics_again": round(1e3 * t_fid2, 2)}, "fid_self_numerical_floor": fid_self}), flush=True)


def cpu_baseline(cfg, sd, steps_total, seconds_budget=20.0):
    """Oracle (CPU port of the reference path) on this host: B=8 motions, a bounded number of the S steps.
    torch's intra-op pool oversubscribes badly on many-core hosts for these small GEMMs, so the thread count
    is calibrated first (the best of a short sweep is used and reported as `cores`)."""
    from oracle import regennet_oracle as orc
    from regennet_amd import synth
    host = os.cpu_count() or 1
    B = 8
    y = {"cmotion": torch.from_numpy(synth.make_cmotion(cfg, B, seed=1))}
    x = torch.from_numpy(synth.make_noise_tape(cfg, B, 0, seed=10)[0])
    t = torch.full((B,), 500, dtype=torch.long)
    best = (None, float("inf"))
    with torch.no_grad():
        for th_n in sorted({c for c in (8, 16, 32, 64, host) if c <= host}):
            torch.set_num_threads(th_n)
            orc.cmdm_forward(sd, cfg, x, t, y)
            t0 = time.perf_counter()
            orc.cmdm_forward(sd, cfg, x, t, y)
            dt1 = time.perf_counter() - t0
            if dt1 < best[1]:
                best = (th_n, dt1)
            if dt1 > 5.0:
                break
        torch.set_num_threads(best[0])
        n, t0 = 0, time.perf_counter()
        while True:
            x0 = orc.cmdm_forward(sd, cfg, x, t, y)
            x = 0.9 * x + 0.1 * x0                          # sampler-update sized elementwise work
            n += 1
            dt = time.perf_counter() - t0
            if dt > seconds_budget or n >= steps_total:
                break
    per_step = dt / n
    out = {"value": B / (per_step * steps_total), "unit": "motions/s", "cores": best[0], "kind": "port",
           "sample": f"oracle (torch-CPU port of the reference path; cross-attention folded, so slightly less work than the reference graph) "
                     f"B={B}, {n} of {steps_total} denoiser evaluations timed ({dt:.1f}s) with {best[0]} threads (best of a sweep; host has "
                     f"{host} logical cores), scaled to a full call"}
    # BASELINE configs[0] exactly where it fits the budget: B=1, the complete sampling loop, same thread count
    if cfg["cond_mode"] == "no_cond" and steps_total <= 1000:
        try:
            sched = orc.make_schedule("cosine", "" if steps_total == 1000 else str(steps_total))
            tape = synth.make_noise_tape(cfg, 1, len(sched[0]), seed=10)
            y1 = {"cmotion": torch.from_numpy(synth.make_cmotion(cfg, 1, seed=1))}
            probe0 = time.perf_counter()
            with torch.no_grad():
                orc.cmdm_forward(sd, cfg, torch.from_numpy(tape[0]), torch.full((1,), 500, dtype=torch.long), y1)
            est = (time.perf_counter() - probe0) * len(sched[0])
            if est < 40.0:
                t0 = time.perf_counter()
                orc.sample_loop(sd, cfg, sched, tape, y1, mode="ddpm")
                dt1 = time.perf_counter() - t0
                out["cfg1_exact"] = {"B": 1, "steps": len(sched[0]), "seconds": round(dt1, 2), "motions_per_s": round(1.0 / dt1, 4)}
        except Exception as e:   # the baseline is informational: never fail the bench line over it
            out["cfg1_exact"] = {"error": repr(e)}
    return out


def self_launch(n, argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks here (one process per GPU over
    RCCL, rendezvous on 127.0.0.1) — the counterpart of the reference's mpi4py bootstrap (utils/dist_util.py:20-42)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC - without it RCCL's intra-node transport fails with
    # `hipIpcGetMemHandle: invalid argument` (the image exports it already; kept explicit for environments that were scrubbed)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed sampling calls")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="motions per GPU")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="STRONG scaling: this many motions in total, sharded over the ranks like cgenerate shards num_samples (dist_util.shard_bounds); "
                         "the line then says \"scaling\": \"strong\" and --batch is ignored (0: weak scaling, --batch motions per GPU)")
    ap.add_argument("--config", default="ntu")
    ap.add_argument("--respacing", default="", help="timestep_respacing ('' = 1000-step DDPM)")
    ap.add_argument("--sampler", default="ddpm", choices=["ddpm", "ddim"])
    ap.add_argument("--guided", action="store_true")
    ap.add_argument("--precision", default=os.environ.get("REGENNET_PRECISION", "bf16_x3tail"),
                    choices=["f32", "bf16x3", "bf16", "bf16_x3tail"])
    ap.add_argument("--x3-tail", type=int, default=None, help="precision schedule: split-bf16 for the last N loop indices")
    ap.add_argument("--f16-steps", type=int, default=None,
                    help="precision schedule: fp16 MFMA operands for the N plain steps in front of the split-bf16 tail (default: the engine's 8 where the one-kernel "
                         "decoder stack runs the plain phase; 0: none, and the bf16 rule's tail)")
    ap.add_argument("--recogniser-f16", action="store_true",
                    help="--config stgcn / eval_pipeline: the recogniser on single fp16 operand planes (SG_F16: one MFMA per product instead of the split-bf16 three)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", default=bool(os.environ.get("REGENNET_FORCE_DIST")),
                    help="take every multi-rank code path with a ONE-rank process group (RCCL init, blob broadcast, barriers, reductions)")
    ap.add_argument("--serial-engine-build", action="store_true", default=bool(os.environ.get("REGENNET_SERIAL_ENGINE_BUILD")),
                    help="ranks repack their checkpoints in turn instead of concurrently")
    ap.add_argument("--no-row-check", action="store_true", help="skip the B = 1 re-run of motions 0 and B-1 of the last timed call")
    ap.add_argument("--profile-evals", type=int, default=3)
    ap.add_argument("--engine-stub", default="", help=argparse.SUPPRESS)   # tests only: "module:Class" replaces _lib.Engine (CPU, gloo);
    a = ap.parse_args(argv)                                                 # the line is then marked "data": "STUB ENGINE ..." and measures nothing
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus, argv))
    if a.config == "stgcn":
        return bench_stgcn(a)
    if a.config == "eval_pipeline":
        return bench_eval_pipeline(a)

    from regennet_amd import synth
    from regennet_amd.model.cfg_sampler import ClassifierFreeSampleModel
    from regennet_amd.utils import dist_util

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # REGENNET_FORCE_DIST=1 (tools): a ONE-rank process group takes every multi-rank code path below - RCCL init, the blob broadcast, the
    # barriers, the MAX all-reduce of the time, the gather of the device names - which is how they are exercised on a 1-GPU box
    if a.force_dist:
        os.environ["REGENNET_FORCE_DIST"] = "1"              # (dist_util.collectives_active reads it)
    multi = world > 1 or a.force_dist
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: refusing to report a line for a different rank count")
    if a.engine_stub:
        from regennet_amd import _lib
        mod, cls = a.engine_stub.split(":")
        _lib.Engine = getattr(importlib.import_module(mod), cls)
    elif torch.cuda.is_available() and a.gpus > torch.cuda.device_count():
        raise SystemExit(f"bench.py: --gpus {a.gpus} but this node has {torch.cuda.device_count()} GPU(s)")
    dev = dist_util.setup_dist()
    assert dev.type == "cuda" or a.engine_stub, "bench.py needs an AMD GPU (no CPU fallback)"
    if multi:
        assert dist.is_initialized() and dist.get_world_size() == a.gpus, "process group does not span --gpus ranks"
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)

    cfg = synth.get_config(a.config)
    B = a.batch
    lo = rank * B                                           # global sample index of this rank's shard (weak scaling: every rank its own B motions)
    if a.global_batch > 0:                                  # strong scaling: contiguous shards of ONE global batch (BASELINE configs[3] / [4]: 1024 / 2048 over 8)
        if a.global_batch < world:
            raise SystemExit(f"bench.py: --global-batch {a.global_batch} < {world} ranks: a rank would have nothing to sample")
        lo, hi = dist_util.shard_bounds(a.global_batch, rank, world)
        B = hi - lo
    # rank 0 owns the checkpoint; other ranks start from a different seed and receive rank 0's parameters via RCCL (dist_util.sync_model_weights)
    sd = synth.make_state_dict(cfg, seed=0 if rank == 0 else 1000 + rank)
    model, diffusion = synth.build_model(cfg, sd, resp=a.respacing, precision=a.precision, device=str(dev), x3_tail=a.x3_tail, f16_steps=a.f16_steps)
    # Engine build = host-side repack of the checkpoint into the device blob (fp32 -> bf16 planes in three layouts) + upload; timed and
    # reported ("engine_build_s": ~2 s of one host core at N = 1). N ranks repack concurrently by default; REGENNET_SERIAL_ENGINE_BUILD=1
    # makes them take turns (a host whose memory bandwidth 8 concurrent repacks would saturate), the broadcast follows either way.
    t_build = time.perf_counter()
    synced_bytes = dist_util.sync_model_weights(model, 0) if multi else 0   # ONE collective over xGMI: rank 0's checkpoint as a flat fp32 buffer
    if multi and a.serial_engine_build:
        eng = None
        for r in range(world):
            if r == rank:
                eng, _ = model._get_engine(B)
            dist.barrier()
    else:
        eng, _ = model._get_engine(B)
    sync()
    build_s = time.perf_counter() - t_build
    fm = ClassifierFreeSampleModel(model) if a.guided else model
    if a.global_batch > 0:     # the conditions of ONE global batch, this rank's slice: with the Philox key = global sample index a motion is the same at any N
        cut = lambda arr: torch.from_numpy(np.ascontiguousarray(arr[lo:lo + B])).to(dev)
        y = {"cmotion": cut(synth.make_cmotion(cfg, a.global_batch, seed=1))}
        if cfg["cond_mode"] == "action":
            y["action"] = cut(synth.make_actions(cfg, a.global_batch, seed=2))
        if cfg["cond_mode"] == "text":
            y["text_features"] = cut(synth.make_text_features(cfg, a.global_batch, seed=3))
    else:
        y = {"cmotion": torch.from_numpy(synth.make_cmotion(cfg, B, seed=1 + rank)).to(dev)}
        if cfg["cond_mode"] == "action":
            y["action"] = torch.from_numpy(synth.make_actions(cfg, B, seed=2 + rank)).to(dev)
        if cfg["cond_mode"] == "text":
            y["text_features"] = torch.from_numpy(synth.make_text_features(cfg, B, seed=3 + rank)).to(dev)
    if a.guided:
        y["scale"] = torch.full((B,), 2.5, device=dev)
    shape = (B, cfg["njoints"], cfg["nfeats"], cfg["num_frames"])
    fn = diffusion.p_sample_loop if a.sampler == "ddpm" else diffusion.ddim_sample_loop
    S = diffusion.num_timesteps

    def one_call(seed):
        return fn(fm, shape, clip_denoised=False, model_kwargs={"y": y}, seed=seed, sample_offset=lo,
                  use_graph=not a.no_graph)

    for w in range(a.warmup):
        out = one_call(10 + w)
    sync()
    if multi:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(a.steps):
        out = one_call(100 + k)
    sync()
    if multi:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    devices = [str(dev)]
    if multi:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        devices = [None] * world
        dist.all_gather_object(devices, f"rank{rank}:{dev}" + (f" ({torch.cuda.get_device_name(dev)})" if dev.type == "cuda" else ""))

    # ---- the timed launch at its real shape: motions 0 and B-1 of the LAST timed call against single-motion runs of the same kernel form
    plan = {} if a.engine_stub else eng.plan_query(B, a.guided, split_phase=False)
    # the precision plan the timed calls followed (rgn_precision_plan: the engine's own answer for this batch on this schedule)
    n16, tail = (0, 0) if a.engine_stub else eng.precision_plan(B, a.guided)
    row_dev = None
    if rank == 0 and not a.engine_stub and not a.no_row_check and a.steps > 0:
        row_dev = row_check(cfg, synth.make_state_dict(cfg, seed=0), a, dev, y, out, 100 + a.steps - 1, lo, plan,
                            "p_sample_loop" if a.sampler == "ddpm" else "ddim_sample_loop", sorted({0, B - 1}))

    # ---- roofline: HIP events around every launch of a short eager single-chain pass on the engine's stream -----------
    # (the first loop indices: the plain-bf16 phase under the precision schedule = where >= 97 % of the evaluations run)
    roof = None
    if rank == 0 and a.profile_evals > 0 and not a.engine_stub:   # (--profile-evals 0: tools/collect_pmc.sh wants the sampling call only)
        eng.profile_enable(True)
        x = torch.empty(shape, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        eng.randn(x, B, 5, lo, st)
        first = S - 1
        n_eval = min(a.profile_evals, S)
        fl = {cls: rec["flops"] for cls, rec in plan.items()}     # the engine's own plan (rgn_plan_query): what it launches, priced by SURVEY 8(d)
        if not fl.get("steps_fused", 0.0) > 0:              # (the multi-step launch was warmed by the timed region; a 1-step launch of it would only
            eng.sample_range(a.sampler, a.guided, 0.0, x, None, 5, lo, first, 1, None, False, False, st)   # skew rocprofv3's average) warm (untimed)
        torch.cuda.synchronize()
        eng.profile_enable(True)                            # reset
        run_steps = 0
        if fl.get("steps_fused", 0.0) > 0:
            # the timed region's dominant launch covers the WHOLE plain-bf16 phase of a call: profile exactly that launch (every
            # k_layers<steps> launch of this command then has the same step count, and rocprofv3's average duration is comparable)
            bulk = S - tail - n16                               # plain-bf16 steps of a call; 0: the schedule's plain steps all run on fp16 operands
            run_steps = n_eval = max(bulk if bulk > 0 else n16, 1)
            eng.randn(x, B, 5, lo, st)
            eng.sample_range(a.sampler, a.guided, 0.0, x, None, 5, lo, first, run_steps, None, False, False, st)
        else:
            eng.sample_range(a.sampler, a.guided, 0.0, x, None, 5, lo, first - 1, n_eval, None, False, False, st)
        torch.cuda.synchronize()
        prof = eng.profile_query()
        ovh_ms = eng.profile_bracket_overhead_ms()
        eng.profile_enable(False)
        peak = PEAK_TFLOPS[a.precision]
        names = {cls: rec["kernel"] for cls, rec in plan.items()}
        per_kernel = []
        for cls, (ms, n) in prof.items():
            if cls == "steps_fused" and run_steps and n >= 1:
                us = max(1e3 * (ms - n * ovh_ms) / n, 0.1)
                tf = fl[cls] * run_steps / (us * 1e-6) / 1e12
                # what the one-kernel form is really bound by: every workgroup (one per sample / motion) streams every layer's weight fragments from its
                # XCD's L2 once per evaluation pass - the L2 -> CU stream, against the guide's aggregate L2 bandwidth (MI355X_MICROARCH.md: 34.5 TB/s)
                l2_bytes = plan[cls]["l2_bytes"]
                l2_tbps = l2_bytes * run_steps / (us * 1e-6) / 1e12
                per_kernel.append({"kernel": names[cls], "launches_per_eval": round(n / run_steps, 6), "steps_per_launch": run_steps, "avg_us": round(us, 2),
                                   "us_per_step": round(us / run_steps, 2), "ms_per_eval": round(us / run_steps * 1e-3, 4), "bound": "mfma",
                                   "algo_gflop_per_launch": round(fl[cls] * run_steps / 1e9, 3), "achieved": round(tf, 1), "peak": peak,
                                   "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                                   "l2_stream": {"bound": "l2", "achieved": round(l2_tbps, 2), "peak": 34.5, "unit": "TB/s", "frac": round(l2_tbps / 34.5, 4),
                                                 "bytes_per_step": round(l2_bytes),
                                                 "note": "weight fragments every workgroup pulls from its XCD's L2 per step (bf16, each layer once per evaluation pass) / "
                                                         "the launch's duration; peak = the guide's aggregate L2 bandwidth at the full clock - at the ~1.95 GHz this kernel "
                                                         "is given the same L2 delivers ~28 TB/s (DESIGN.md 4.0d)"}})
                continue
            if n < n_eval or cls not in names:        # (classes that ran once per call, e.g. the embedding in front of the first fused step)
                continue
            # every event bracket carries the dispatch + event latency of an empty bracket (calibrated on a no-op kernel,
            # whose own ~2 us of execution stay in the figure): subtract it, leaving ~ the rocprofv3 kernel duration
            us = max(1e3 * (ms - n * ovh_ms) / n + 2.0, 0.1)
            e = {"kernel": names[cls], "launches_per_eval": n // n_eval, "avg_us": round(us, 2),
                 "ms_per_eval": round(us * (n // n_eval) * 1e-3, 4)}
            if fl.get(cls, 0.0) > 0:
                tf = fl[cls] * n_eval / n / (us * 1e-6) / 1e12
                e.update(bound="mfma", algo_gflop_per_launch=round(fl[cls] * n_eval / n / 1e9, 3), achieved=round(tf, 1),
                         peak=peak, unit="TFLOP/s", frac=round(tf / peak, 4))
            per_kernel.append(e)
        per_kernel.sort(key=lambda e: -e["ms_per_eval"])
        dom = next(e for e in per_kernel if "frac" in e)     # the class with the largest share of the step that does MFMA work
        phase = ""
        if a.precision == "bf16_x3tail":
            phase = " (plain phase of the precision schedule: single-plane operands, one MFMA per product)"
        roof = {"bound": "mfma", "kernel": dom["kernel"] + phase, "achieved": dom["achieved"], "peak": peak, "unit": "TFLOP/s",
                "frac": dom["frac"], "traffic": None,
                "avg_launch_us": dom["avg_us"], "launches_per_eval": dom["launches_per_eval"], "steps_per_launch": dom.get("steps_per_launch"),
                "algo_gflop_per_launch": dom["algo_gflop_per_launch"],
                "event_bracket_overhead_us": round(1e3 * ovh_ms, 2),
                "sustained_peak_note": "a pure v_mfma_f32_32x32x16_bf16 loop on all 1024 SIMDs sustains 1826 TFLOP/s on this pool (1.97 GHz at "
                                       "1320 W; the sampling loop itself runs at 1.98 GHz / 1230 W): DESIGN.md 10.3. peak/frac use the guide's 2500",
                "note": "achieved = algorithmic FLOPs of the class's launches / their summed duration; duration = HIP-event "
                        "bracket on the engine stream minus the calibrated empty-bracket latency (single-chain eager pass; "
                        "matches rocprofv3 --kernel-trace durations, profiles/). k_layers<steps>: ONE launch carries every sample through "
                        "steps_per_launch complete sampler steps (decoder stack + step boundary), exactly the launch the timed region issues "
                        "once per call; other forms: the timed region replays multi-chain hipGraphs in which kernels of different chains overlap",
                "per_kernel": per_kernel}
        if dom.get("l2_stream"):
            roof["l2_stream"] = dom["l2_stream"]
        # HBM bytes per launch of the dominant kernel: NOT measured by this run (counters need rocprofv3 passes of their own, the
        # MI355X guide's recipe) - read from the newest committed PMC summary (tools/collect_pmc.sh + tools/summarize_pmc.py) and
        # labelled as such
        key = f"{a.config}_B{B}_{a.precision}_{'cfg' if a.guided else 'plain'}"
        for name in ("r06_pmc_bench.json", "r05_pmc_bench.json", "r04_pmc_bench.json", "r03_pmc_bench.json", "r02_pmc_bench.json"):
            pmc = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc):
                with open(pmc) as fh:
                    per = json.load(fh).get(key, {})
                # the PMC summary names kernel CLASSES (tools/summarize_pmc.py), the plan names instantiations
                alias = next((name for sub, name in (("k_layers<true", "k_layers<steps>"), ("k_layers", "k_layers"), ("k_step", "k_step"), ("k_mlp", "k_mlp"),
                                                     ("k_qkv_attn_long", "k_qkv_attn_long"), ("k_qkv_attn", "k_qkv_attn"), ("k_sb_gemm", "k_sb_gemm"),
                                                     ("k_gemm_x3", "k_gemm_x3")) if sub in dom["kernel"]), dom["kernel"])
                rec = per.get(dom["kernel"]) or per.get(alias) or {}
                if rec.get("hbm_bytes_per_launch"):
                    roof["traffic"] = rec["hbm_bytes_per_launch"]
                    counted = "counted on a launch of the same step count"
                    if rec.get("steps_per_launch") and dom.get("steps_per_launch"):
                        roof["traffic_per_step"] = round(rec["hbm_bytes_per_launch"] / rec["steps_per_launch"])
                        if rec["steps_per_launch"] != dom["steps_per_launch"]:   # counted on a launch of another length: scale to the timed launch's step count
                            roof["traffic"] = round(rec["hbm_bytes_per_launch"] / rec["steps_per_launch"] * dom["steps_per_launch"])
                            counted = f"counted on a {rec['steps_per_launch']}-step launch and scaled to {dom['steps_per_launch']} steps"
                    roof["traffic_source"] = (f"profiles/{name} [{key}][{alias}]: FETCH_SIZE x 2 (gfx950) + WRITE_SIZE from separate "
                                              f"rocprofv3 --pmc passes of a committed earlier run of this command ({counted}); not measured in this run")
                    roof["mfma_util_pmc"] = rec.get("mfma_util")
                    break

    if rank == 0:
        evals = S * (2 if a.guided else 1)
        algo = ALGO_GFLOP_PER_EVAL.get((cfg["dataset"], cfg["cm_mode"]), None)
        total = a.global_batch if a.global_batch > 0 else B * world
        value = a.steps * total / dt
        dtype = a.precision
        if a.precision == "bf16_x3tail":
            dtype = (f"bf16 MFMA operands, fp32 accumulate/LayerNorm/softmax, for {S - tail - n16} of {S} steps; " +
                     (f"fp16 MFMA operands for the next {n16}; " if n16 else "") + f"split-bf16 (x3) for the last {tail}")
        line = {
            "metric": "sampled motions/sec", "value": round(value, 3), "unit": "motions/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 2), "higher_is_better": True,
            "scaling": "strong" if a.global_batch > 0 else "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic" if not a.engine_stub else f"STUB ENGINE {a.engine_stub}: launcher test, nothing measured",
            "engine_build_s": round(build_s, 2), "weights_broadcast_bytes": synced_bytes,
            "headline_row_check_max_abs": row_dev,
            "headline_row_check": None if row_dev is None else f"motions 0 and {B - 1} of the last timed call vs B = 1 runs of the same kernel form "
                                                               f"with the motion's global Philox key (bit-exact in the plain-bf16 phase; <= 2e-5 behind the split-bf16 tail)",
            "rccl_world_size": dist.get_world_size() if dist.is_initialized() else 1,
            "backend": dist.get_backend() if dist.is_initialized() else None, "devices": devices,
            "config": {"workload": f"{a.config}: [B={B}/GPU,56,6,{cfg['num_frames']}] online/{cfg['cm_mode']}/{cfg['cond_mode']} "
                                   f"L{cfg['layers']} d{cfg['latent_dim']}, {S}-step {a.sampler.upper()}"
                                   f"{' + CFG 2.5' if a.guided else ''}, Philox noise, hipGraph={'off' if a.no_graph else 'on'}",
                       "batch_per_gpu": B if a.global_batch <= 0 else f"{a.global_batch // world}..{-(-a.global_batch // world)}", "global_batch": total,
                       "denoiser_evals_per_motion": evals,
                       "parallelism": f"batch-shard x{world}"},
        }
        if algo is not None:
            e2e = value * evals * algo / 1e3
            line["e2e_algorithmic_tflops"] = round(e2e, 2)
            line["e2e_frac_of_peak"] = round(e2e / (PEAK_TFLOPS[a.precision] * world), 4)
        line["roofline"] = roof
        if world == 1 and not a.no_cpu_baseline and not a.engine_stub:
            line["cpu_baseline"] = cpu_baseline(cfg, synth.make_state_dict(cfg, seed=0), evals)
            line["speedup_vs_cpu_baseline"] = round(value / line["cpu_baseline"]["value"], 1)
    # The JSON line must be the LAST thing on stdout: RCCL prints a version banner through C stdio, which is flushed when the process
    # exits - i.e. behind a line printed here. So: every rank flushes C stdio, the ranks meet, rank 0 prints the line, and from then
    # on every rank's stdout goes to stderr.
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if multi:
        dist.barrier()
    if rank == 0:
        print(json.dumps(line), flush=True)
    os.dup2(2, 1)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
