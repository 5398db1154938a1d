# PMC passes over the recogniser bench (run on the GPU box): what bounds its GEMMs? -> gpurun_out/pmc_stgcn/summary.json
# PMC_STGCN_FLAGS="--recogniser-f16": the fp16 form (its kernels are the <..., true> instantiations; the bench's comparison forward adds the default's once)
set -u
FLAGS=${PMC_STGCN_FLAGS:-}
R=${GRAFT_REPO_ROOT:-$PWD}
rm -rf "$R/gpurun_out/pmc_stgcn"; mkdir -p "$R/gpurun_out/pmc_stgcn"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$R/gpurun_out/pmc_stgcn" -o "pass$i" -- python "$R/bench.py" --config stgcn $FLAGS --steps 1 --warmup 0 > "$R/gpurun_out/pmc_stgcn/pass$i.log" 2>&1 < /dev/null
    echo "pass $i ($grp): rc=$?"
done
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob("$R/gpurun_out/pmc_stgcn/**/*counter_collection.csv", recursive=True)):
    tot, dur = collections.defaultdict(float), {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        tot[(r["Dispatch_Id"], k, r["Counter_Name"])] += float(r["Counter_Value"])
        dur[(r["Dispatch_Id"], k)] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for (d, k, c), v in tot.items():
        acc[k][c].append(v)
        if c == "SQ_VALU_MFMA_BUSY_CYCLES": acc[k]["_dur"].append(dur[(d, k)])
out = {}
for k, cs in acc.items():
    m = {c: sum(v) for c, v in cs.items()}     # totals over the forward passes of the run (60 + 150 frames)
    e = {"launches": len(next(iter(cs.values())))}
    if "FETCH_SIZE" in m: e["hbm_fetch_GB"] = round(m["FETCH_SIZE"] * 2048 / 1e9, 2)
    if "WRITE_SIZE" in m: e["hbm_write_GB"] = round(m["WRITE_SIZE"] * 1024 / 1e9, 2)
    if "TCC_HIT_sum" in m: e["l2_hit"] = round(m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 3)
    if "_dur" in m and m["_dur"] > 0: e["mfma_util"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * 2.4e9 * m["_dur"] * 1e-9), 3); e["ms_total"] = round(m["_dur"] / 1e6, 2)
    if "SQ_WAIT_INST_ANY" in m: e["issue_stall"] = round(m["SQ_WAIT_INST_ANY"] / max(1.0, m["SQ_WAVE_CYCLES"]), 3)
    if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" in m: e["wait_any"] = round(m["SQ_WAIT_ANY"] / max(1.0, m["SQ_WAVE_CYCLES"]), 3)
    out[k] = e
json.dump(out, open("$R/gpurun_out/pmc_stgcn/summary.json", "w"), indent=1)
for k, e in sorted(out.items(), key=lambda kv: -kv[1].get("ms_total", 0)): print(k, e)
PY
