# round 5: graph aggregation fused into the 1 x 1 convolution's operand fragments (k_sg_gcn) against the two-launch form, same box
mkdir -p gpurun_out/r05e
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_eval_gpu.py -x -q > gpurun_out/r05e/eval_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05e/eval_tests.log
tail -5 gpurun_out/r05e/eval_tests.log
for rep in 1 2; do
  echo "fused:"; timeout 300 python bench.py --config stgcn --steps 10 --warmup 2 2>/dev/null | tee gpurun_out/r05e/bench_fused_$rep.json | cut -c1-200
  echo "fused<=128:"; REGENNET_SG_GCN_BN=128 timeout 300 python bench.py --config stgcn --steps 10 --warmup 2 2>/dev/null | tee gpurun_out/r05e/bench_fused128_$rep.json | cut -c1-200
  echo "two launches:"; REGENNET_SG_NO_GCN_FUSE=1 timeout 300 python bench.py --config stgcn --steps 10 --warmup 2 2>/dev/null | tee gpurun_out/r05e/bench_unfused_$rep.json | cut -c1-200
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05e/prof -o stgcn -- python $GRAFT_REPO_ROOT/bench.py --config stgcn --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r05e/stgcn_prof.log 2>&1)
find gpurun_out/r05e/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05e/stgcn_kernel_stats.csv \;
rm -rf gpurun_out/r05e/prof
head -14 gpurun_out/r05e/stgcn_kernel_stats.csv | cut -c1-60,150-330
