# round 5: where k_sg_tconv's k-step goes - timing builds with one part removed (results wrong), same box; kernel stats per build
cp regennet_amd/libregennet_hip.so /tmp/lib_keep.so
export TMPDIR=/tmp
for L in "$@"; do
  cp build/lib_$L.so regennet_amd/libregennet_hip.so
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$L -o s -- python $GRAFT_REPO_ROOT/bench.py --config stgcn --steps 3 --warmup 1 > /dev/null 2>&1)
  echo "== $L"; python - <<PY
import csv, glob
f = glob.glob("/tmp/prof_$L/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_sg_tconv<" in r["Name"]: print("  ", r["Name"].split("(")[0][-28:], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
cp /tmp/lib_keep.so regennet_amd/libregennet_hip.so
