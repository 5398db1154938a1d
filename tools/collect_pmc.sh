#!/bin/bash
# Per-kernel hardware counters of one short sampling call in the plain-bf16 phase (run on the GPU box):
#   tools/collect_pmc.sh [out.json] [workload-key] [extra bench.py flags...]
# writes the raw CSVs under gpurun_out/pmc/ and a per-kernel summary JSON (tools/summarize_pmc.py).
# One rocprofv3 pass per counter group (FETCH_SIZE and WRITE_SIZE each alone, as the MI355X guide prescribes; never
# combined with trace domains). Single kernel chain (REGENNET_STREAMS=1) and eager launches so every kernel is a plain
# dispatch the counters attach to. Each pass is bounded by `timeout`.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=${1:-gpurun_out/pmc_summary.json}
KEY=${2:-}
shift 2 2>/dev/null || true
case "$OUT" in /*) ;; *) OUT="$R/$OUT" ;; esac
rm -rf "$R/gpurun_out/pmc"
mkdir -p "$R/gpurun_out/pmc"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    # PMC_FULL=1: the timed region's own launch - the default schedule (1000-step DDPM: ONE k_layers<steps> dispatch of 995 steps per call) instead
    # of a 3-step schedule, so that roofline.traffic is COUNTED on the launch bench.py times, not scaled from a short one
    if [ -n "${PMC_FULL:-}" ]; then SCHED=""; else SCHED="--respacing 3 --x3-tail 0"; fi
    REGENNET_STREAMS=1 timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$R/gpurun_out/pmc" -o "pass$i" -- \
        python "$R/bench.py" $SCHED --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-row-check --profile-evals 0 "$@" \
        > "$R/gpurun_out/pmc/pass$i.log" 2>&1 < /dev/null
    echo "pass $i ($grp): rc=$?"
done
python "$R/tools/summarize_pmc.py" "$R/gpurun_out/pmc" "$OUT" $KEY ${PMC_STEPS:-3}   # (3-step schedules: a k_layers<steps> dispatch covers 3 steps; PMC_FULL: PMC_STEPS=995)
