"""Summarise rocprofv3 --pmc CSVs (tools/collect_pmc.sh) into per-kernel means per launch.

    python tools/summarize_pmc.py <dir with *counter_collection.csv> <out.json> [workload-key] [steps-per-launch]

(steps-per-launch: how many sampler steps ONE k_layers<steps> dispatch of the counted run covered - tools/collect_pmc.sh runs 3-step
schedules - so that bench.py can scale the per-launch bytes to the launch it times.)

Units (MI355X_MICROARCH.md, counter rows): FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies 64 B per 128-B
request, so fetched bytes = FETCH_SIZE * 1024 * 2. SQ_VALU_MFMA_BUSY_CYCLES counts CYCLES, summed over every SIMD of the
chip (= 32 x the number of 32x32x16 bf16 MFMAs issued); SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / SQ_WAIT_* count QUAD-cycles, so
their ratio to the MFMA counter means nothing. The matrix-pipe utilisation reported here is
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x 2.4 GHz x dispatch duration)
with the duration from the dispatch's own Start/End timestamps in the counter CSV, i.e. busy matrix-pipe cycles against
the cycles the pipes offer at the peak clock — the same convention as a roofline fraction against the 2.5 PFLOP/s dense
bf16 peak (for a kernel that issues no redundant MFMAs the two agree). Issue-stall share = SQ_WAIT_INST_ANY /
SQ_WAVE_CYCLES (both quad-cycles). Derived keys: hbm_fetch_MB_per_launch_corrected, hbm_write_MB_per_launch,
hbm_bytes_per_launch (what bench.py reports as roofline.traffic), l2_hit_rate, mfma_util, issue_stall_frac.
"""
import collections
import csv
import glob
import json
import os
import sys

# substring of the kernel name -> the name bench.py uses (first match wins)
CLASSES = [("k_layers<true, false, true>", "k_layers<steps, f16>"), ("k_layers<true, true, true>", "k_layers<steps, f16>"),   # the fp16 sub-phase's short launch: its own class
           ("k_layers<true", "k_layers<steps>"), ("k_layers", "k_layers"), ("k_step", "k_step"), ("k_mlp", "k_mlp"), ("k_rowgemm<0", "k_rowgemm<LN>"), ("k_rowgemm<1", "k_rowgemm<ACT>"), ("k_gemm_x3", "k_gemm_x3"),
           ("k_qkv_attn_long", "k_qkv_attn_long"), ("k_qkv_attn", "k_qkv_attn"), ("k_attn_x3", "k_attn_x3"), ("k_sb_gemm", "k_sb_gemm"), ("k_layernorm", "k_layernorm"), ("k_update", "k_update"),
           ("k_gemm_bf16", "k_gemm_bf16"), ("k_gemm_f32", "k_gemm_f32"), ("k_attn_mfma", "k_attn_mfma")]
N_SIMD = 1024          # 256 CUs x 4
PEAK_CLOCK_HZ = 2.4e9


def main(src, out, key=None, steps_per_launch=None):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
        tot, dur = collections.defaultdict(float), {}
        with open(path) as f:
            for r in csv.DictReader(f):
                tot[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
                dur[(r["Dispatch_Id"], r["Kernel_Name"])] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        for (did, kname, cname), v in tot.items():
            for sub, name in CLASSES:
                if sub in kname:
                    acc[name][cname].append(v)
                    if cname == "SQ_VALU_MFMA_BUSY_CYCLES":
                        acc[name]["_mfma_pass_duration_ns"].append(dur[(did, kname)])
                    break
    summary = {k: {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in sorted(cs.items())} for k, cs in acc.items()}
    for k, cs in summary.items():
        m = {c: s["mean"] for c, s in cs.items()}
        if "FETCH_SIZE" in m:
            cs["hbm_fetch_MB_per_launch_corrected"] = round(m["FETCH_SIZE"] * 1024 * 2 / 1e6, 2)
        if "WRITE_SIZE" in m:
            cs["hbm_write_MB_per_launch"] = round(m["WRITE_SIZE"] * 1024 / 1e6, 2)
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            cs["hbm_bytes_per_launch"] = round(m["FETCH_SIZE"] * 1024 * 2 + m["WRITE_SIZE"] * 1024)
        if "TCC_HIT_sum" in m and m["TCC_HIT_sum"] + m.get("TCC_MISS_sum", 0) > 0:
            cs["l2_hit_rate"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("_mfma_pass_duration_ns", 0) > 0:
            cs["mfma_util"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * PEAK_CLOCK_HZ * m["_mfma_pass_duration_ns"] * 1e-9), 4)
        if "SQ_WAIT_INST_ANY" in m and m.get("SQ_WAVE_CYCLES", 0) > 0:
            cs["issue_stall_frac"] = round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 3)
    if steps_per_launch and "k_layers<steps>" in summary:
        summary["k_layers<steps>"]["steps_per_launch"] = int(steps_per_launch)
    result = {key: summary} if key else summary
    if key and os.path.exists(out):          # one file holds several workloads
        with open(out) as f:
            old = json.load(f)
        old.update(result)
        result = old
    with open(out, "w") as f:
        json.dump(result, f, indent=1)
    for k, cs in summary.items():
        print(k, {c: (round(s["mean"], 1) if isinstance(s, dict) else s) for c, s in cs.items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None, sys.argv[4] if len(sys.argv) > 4 else None)
