# round 5: the resident-window temporal convolution (k_sg_tconv) against the row-shifted GEMM (REGENNET_SG_NO_WINDOW=1), same box: parity tests, then bench lines
mkdir -p gpurun_out/r05d
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_eval_gpu.py -x -q > gpurun_out/r05d/eval_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05d/eval_tests.log
tail -5 gpurun_out/r05d/eval_tests.log
for rep in 1 2; do
  echo "window128:"; timeout 300 python bench.py --config stgcn --steps 10 --warmup 2 2>/dev/null | tee gpurun_out/r05d/bench_w128_$rep.json | cut -c1-200
  echo "shifted:"; REGENNET_SG_NO_WINDOW=1 timeout 300 python bench.py --config stgcn --steps 10 --warmup 2 2>/dev/null | tee gpurun_out/r05d/bench_shift_$rep.json | cut -c1-200
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05d/prof -o stgcn -- python $GRAFT_REPO_ROOT/bench.py --config stgcn --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r05d/stgcn_prof.log 2>&1)
find gpurun_out/r05d/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05d/stgcn_kernel_stats.csv \;
rm -rf gpurun_out/r05d/prof
head -12 gpurun_out/r05d/stgcn_kernel_stats.csv | cut -c1-180
