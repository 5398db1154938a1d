# Same-box A/B of ONE kernel-selection switch (a REGENNET_<KEY> environment variable) under bench.py, optionally behind a parity-test file and followed by
# rocprofv3 kernel stats of both settings:
#     tools/ab_switch.sh [--tests tests/test_eval_gpu.py] [--stats] [--rounds N] REGENNET_SG_NO_WINDOW[=1] -- --config stgcn --steps 10 --warmup 2
# prints "<KEY>=<unset|value>  value  ms_per_step  [per-length ms]" per run. (The one-off r05_* scripts of round 5 - tconv window, fused aggregation, fused
# tail, tile shapes, stride-2 windows, polyphase tail - were this script with the switch and the output directory spelled out; their results are
# profiles/r05/stgcn_*.txt. A -DRGN_SG_PROF build + tools/sg_stamps.py gives the in-kernel cycle stamps.)
set -u
export TMPDIR=/tmp
TESTS=""; STATS=0; N=2
while [ $# -gt 0 ]; do case "$1" in --tests) TESTS=$2; shift 2;; --stats) STATS=1; shift;; --rounds) N=$2; shift 2;; *) break;; esac; done
SW=$1; shift; [ "${1:-}" = "--" ] && shift
KEY=${SW%%=*}; VAL=1; [ "$SW" != "$KEY" ] && VAL=${SW#*=}
[ -n "$TESTS" ] && timeout 1200 python -m pytest $TESTS -x -q 2>&1 | tail -2
line() { python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  $1', d['value'], d['ms_per_step'], [p['ms_per_forward'] for p in d.get('per_length', [])])"; }
for r in $(seq $N); do
  unset $KEY; timeout 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null | line "$KEY=unset"
  export $KEY=$VAL; timeout 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null | line "$KEY=$VAL"
done
if [ $STATS = 1 ]; then
  R=$PWD
  for v in unset $VAL; do
    if [ $v = unset ]; then unset $KEY; else export $KEY=$v; fi
    (cd /tmp && rm -rf /tmp/abs_prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abs_prof -o s -- python $R/bench.py --no-cpu-baseline "$@" > /dev/null 2>&1)
    python - <<PY
import csv, glob
f = glob.glob("/tmp/abs_prof/**/*kernel_stats.csv", recursive=True)
print("== kernel stats, $KEY=$v")
for r in (list(csv.DictReader(open(f[0])))[:14] if f else []):
    print("   ", r["Name"].split("(")[0][-45:].ljust(46), r["Calls"].rjust(5), f"{float(r['TotalDurationNs']) / 1e6:8.2f} ms", f"{float(r['AverageNs']) / 1e3:9.1f} us", r["Percentage"])
PY
  done
fi
