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
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# minimal algorithmic work of ONE denoiser evaluation for ONE sample (FLOP = 2*MAC)
ALGO_GFLOP_PER_EVAL = {("ntu", "concat"): 2.154, ("ntu", "add"): 2.123, ("chi3d", "concat"): 5.593, ("chi3d", "add"): 5.514}
PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0}   # MI355X_MICROARCH.md dense MFMA peaks


def fused_qkv_attention(cfg, precision):
    """Mirrors qkv_attn_supported() in rgn_qkv_attn.hip: the in_proj GEMM and the attention run as one kernel."""
    return (precision != "f32" and cfg["num_frames"] + int(bool(cfg.get("emb_trans_dec"))) <= 64
            and cfg["latent_dim"] // cfg["num_heads"] == 128 and not os.environ.get("REGENNET_NO_FUSED_QKV"))


def gemm_flops_per_eval(cfg, B, guided, precision="bf16x3"):
    """Algorithmic FLOPs of the GEMM-class launches of one evaluation (everything except attention scores/AV; the
    timestep MLP and the folded 1-token cross-attention are per-schedule work, not per-step)."""
    T, d, ff, L, F = cfg["num_frames"], cfg["latent_dim"], cfg["ff_size"], cfg["layers"], cfg["njoints"] * cfg["nfeats"]
    Bm = 2 * B if guided else B
    M = Bm * T
    mac = M * (3 * d * d + d * d + 2 * d * ff) * L          # qkv, out_proj, ffn1, ffn2
    mac += B * T * F * d + M * d * F                        # input embedding (folded fuse half), output projection
    if fused_qkv_attention(cfg, precision):                 # k_qkv_attn carries the (full T x T) scores + AV work too
        mac += M * 2 * T * d * L
    return 2.0 * mac


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
    return {"value": B / (per_step * steps_total), "unit": "motions/s", "cores": best[0], "kind": "port",
            "sample": f"oracle (torch-CPU port of the reference path) B={B}, {n} of {steps_total} denoiser evaluations timed "
                      f"({dt:.1f}s) with {best[0]} threads (best of a sweep; host has {host} logical cores), scaled to a full call"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed sampling calls")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="motions per GPU")
    ap.add_argument("--config", default="ntu")
    ap.add_argument("--respacing", default="", help="timestep_respacing ('' = 1000-step DDPM)")
    ap.add_argument("--sampler", default="ddpm", choices=["ddpm", "ddim"])
    ap.add_argument("--guided", action="store_true")
    ap.add_argument("--precision", default=os.environ.get("REGENNET_PRECISION", "bf16x3"), choices=["f32", "bf16x3", "bf16"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-evals", type=int, default=3)
    a = ap.parse_args()

    from regennet_amd import synth
    from regennet_amd.model.cfg_sampler import ClassifierFreeSampleModel
    from regennet_amd.utils import dist_util
    from tests.helpers import build_hip

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dev = dist_util.setup_dist()
    assert dev.type == "cuda", "bench.py needs an AMD GPU (no CPU fallback)"

    cfg = synth.get_config(a.config)
    B = a.batch
    # rank 0 owns the checkpoint; other ranks start from a different seed and receive the packed blob via RCCL
    sd = synth.make_state_dict(cfg, seed=0 if rank == 0 else 1000 + rank)
    model, diffusion = build_hip(cfg, sd, resp=a.respacing, precision=a.precision, device=str(dev))
    eng, _ = model._get_engine(B)
    if world > 1:
        ptr, nbytes = eng.weight_blob()

        blob = dist_util.device_view(ptr, nbytes, dev)      # zero-copy uint8 view of the packed weight blob
        dist_util.broadcast_flat(blob, 0)                   # ONE collective over xGMI
        torch.cuda.synchronize()
    fm = ClassifierFreeSampleModel(model) if a.guided else model
    lo = rank * B                                           # global sample index of this rank's shard
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
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        out = one_call(100 + k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- roofline of the dominant kernel class: HIP events around every launch, eager mode -----------------
    roof = None
    if rank == 0 and a.profile_evals > 0:   # (--profile-evals 0: tools/collect_pmc.sh wants the sampling call only)
        eng.profile_enable(True)
        x = torch.empty(shape, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        eng.randn(x, B, 5, lo, st)
        first = S - 1
        eng.profile_enable(True)                            # reset after the randn launch
        eng.sample_range(a.sampler, a.guided, 0.0, x, None, 5, lo, first, min(a.profile_evals, S), None, False, False, st)
        torch.cuda.synchronize()
        prof = eng.profile_query()
        eng.profile_enable(False)
        n_eval = min(a.profile_evals, S)
        gemm_ms, gemm_n = prof["gemm_mfma"]
        fl = gemm_flops_per_eval(cfg, B, a.guided, a.precision) * n_eval
        achieved = fl / (gemm_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[a.precision]
        traffic = None   # HBM bytes per GEMM launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_bench_streams1.json")   # FETCH x2 gfx950 correction), same workload only
        if os.path.exists(pmc) and (a.config, B, a.precision, a.guided) == ("ntu", 256, "bf16x3", False):
            with open(pmc) as fh:
                pj = json.load(fh)
            n1, n2 = pj["k_gemm_x3"]["FETCH_SIZE"]["launches"], pj["k_qkv_attn"]["FETCH_SIZE"]["launches"]
            t1 = pj["k_gemm_x3"]["hbm_fetch_MB_per_launch_corrected"] + pj["k_gemm_x3"]["hbm_write_MB_per_launch"]
            t2 = pj["k_qkv_attn"]["hbm_fetch_MB_per_launch_corrected"] + pj["k_qkv_attn"]["hbm_write_MB_per_launch"]
            traffic = round((n1 * t1 + n2 * t2) / (n1 + n2) * 1e6)
        roof = {"bound": "mfma", "kernel": "k_gemm_x3 / k_qkv_attn (all MFMA GEMM launches of one denoiser evaluation)",
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "traffic": traffic,
                "note": "HIP events on the engine stream around every GEMM launch, single-chain eager pass; each bracket carries "
                        "~8-10 us of dispatch/event latency that rocprofv3's kernel timestamps do not (profiles/*_streams1.csv); "
                        "the timed region replays a 4-chain hipGraph in which kernels of different chains overlap",
                "launches_per_eval": gemm_n // n_eval, "avg_launch_us": round(1e3 * gemm_ms / max(gemm_n, 1), 2),
                "class_ms_per_eval": {k: round(v[0] / n_eval, 4) for k, v in prof.items()}}

    if rank == 0:
        evals = S * (2 if a.guided else 1)
        algo = ALGO_GFLOP_PER_EVAL.get((cfg["dataset"], cfg["cm_mode"]), None)
        value = a.steps * B * world / dt
        line = {
            "metric": "sampled motions/sec", "value": round(value, 3), "unit": "motions/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"{a.config}: [B={B}/GPU,56,6,{cfg['num_frames']}] online/{cfg['cm_mode']}/{cfg['cond_mode']} "
                                   f"L{cfg['layers']} d{cfg['latent_dim']}, {S}-step {a.sampler.upper()}"
                                   f"{' + CFG 2.5' if a.guided else ''}, Philox noise, hipGraph={'off' if a.no_graph else 'on'}",
                       "batch_per_gpu": B, "global_batch": B * world, "denoiser_evals_per_motion": evals,
                       "parallelism": f"batch-shard x{world}"},
        }
        if algo is not None:
            e2e = value * evals * algo / 1e3
            line["e2e_algorithmic_tflops"] = round(e2e, 2)
            line["e2e_frac_of_peak"] = round(e2e / (PEAK_TFLOPS[a.precision] * world), 4)
        line["roofline"] = roof
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, synth.make_state_dict(cfg, seed=0), evals)
            line["speedup_vs_cpu_baseline"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
