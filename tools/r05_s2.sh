# round 5: the stride-2 blocks on two resident windows + shortcut k-steps + fused tail (k_sg_tconv_s2) against row-shifted GEMM + shortcut GEMM + k_sg_post, same box
mkdir -p gpurun_out/r05h
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_eval_gpu.py -x -q > gpurun_out/r05h/eval_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05h/eval_tests.log
tail -5 gpurun_out/r05h/eval_tests.log
for rep in 1 2; do
  echo "s2 window:"; timeout 300 python bench.py --config stgcn --steps 10 --warmup 2 2>/dev/null | tee gpurun_out/r05h/bench_s2_$rep.json | cut -c1-200
  echo "generic:"; REGENNET_SG_NO_S2_WINDOW=1 timeout 300 python bench.py --config stgcn --steps 10 --warmup 2 2>/dev/null | tee gpurun_out/r05h/bench_generic_$rep.json | cut -c1-200
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05h/prof -o stgcn -- python $GRAFT_REPO_ROOT/bench.py --config stgcn --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r05h/stgcn_prof.log 2>&1)
find gpurun_out/r05h/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05h/stgcn_kernel_stats.csv \;
rm -rf gpurun_out/r05h/prof
python - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/r05h/stgcn_kernel_stats.csv')))[:15]:
    print(r['Name'].split('(')[0][-45:].ljust(46), r['Calls'].rjust(5), f"{float(r['TotalDurationNs'])/1e6:8.2f} ms", f"{float(r['AverageNs'])/1e3:9.1f} us", r['Percentage'])
PY
