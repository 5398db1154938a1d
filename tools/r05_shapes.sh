# round 5: k_sg_tconv tile shapes (512-row / 256-wide tiles against 256 x <= 128), same box
mkdir -p gpurun_out/r05g
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_eval_gpu.py -x -q > gpurun_out/r05g/eval_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05g/eval_tests.log
tail -5 gpurun_out/r05g/eval_tests.log
for rep in 1 2; do
  echo "big:"; timeout 300 python bench.py --config stgcn --steps 10 --warmup 2 2>/dev/null | tee gpurun_out/r05g/bench_big_$rep.json | cut -c1-200
  echo "small:"; REGENNET_SG_TCONV_SMALL=1 timeout 300 python bench.py --config stgcn --steps 10 --warmup 2 2>/dev/null | tee gpurun_out/r05g/bench_small_$rep.json | cut -c1-200
done
for v in big small; do
if [ $v = small ]; then export REGENNET_SG_TCONV_SMALL=1; fi
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05g/prof -o stgcn -- python $GRAFT_REPO_ROOT/bench.py --config stgcn --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r05g/stgcn_prof.log 2>&1)
find gpurun_out/r05g/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05g/stgcn_kernel_stats_$v.csv \;
rm -rf gpurun_out/r05g/prof
python - <<PY
import csv
print("== $v")
for r in list(csv.DictReader(open('gpurun_out/r05g/stgcn_kernel_stats_$v.csv')))[:13]:
    print(r['Name'].split('(')[0][-45:].ljust(46), r['Calls'].rjust(5), f"{float(r['TotalDurationNs'])/1e6:8.2f} ms", f"{float(r['AverageNs'])/1e3:9.1f} us", r['Percentage'])
PY
done
