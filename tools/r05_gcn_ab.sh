# round 5: two builds of k_sg_gcn on one box (library files given as arguments): parity tests of the first, then k_sg_gcn kernel times of each
export TMPDIR=/tmp
cp regennet_amd/libregennet_hip.so /tmp/lib_keep.so
for L in "$@"; do
  cp build/lib_$L.so regennet_amd/libregennet_hip.so
  timeout 600 python -m pytest tests/test_eval_gpu.py -x -q 2>&1 | tail -1
  for rep in 1 2; do timeout 300 python bench.py --config stgcn --steps 10 --warmup 2 2>/dev/null | python -c "import sys, json; print('  $L ms/forward', json.loads(sys.stdin.readline())['ms_per_step'])"; done
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$L -o s -- python $GRAFT_REPO_ROOT/bench.py --config stgcn --steps 3 --warmup 1 > /dev/null 2>&1)
  python - <<PY
import csv, glob
f = glob.glob("/tmp/prof_$L/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_sg_gcn" in r["Name"]: print("    ", r["Name"].split("(")[0][-18:], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
cp /tmp/lib_keep.so regennet_amd/libregennet_hip.so
