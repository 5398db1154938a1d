# round 5: the split-bf16 MFMA build of the ST-GCN evaluator - parity tests, bench line, rocprofv3 kernel stats
mkdir -p gpurun_out/r05c
export TMPDIR=/tmp
python -m pytest tests/test_eval_gpu.py -x -q -s > gpurun_out/r05c/eval_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05c/eval_tests.log
python bench.py --config stgcn --steps 10 --warmup 2 > gpurun_out/r05c/bench_stgcn.json 2> gpurun_out/r05c/bench_stgcn.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05c/prof -o stgcn -- python $GRAFT_REPO_ROOT/bench.py --config stgcn --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r05c/stgcn_prof.log 2>&1)
find gpurun_out/r05c/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05c/stgcn_kernel_stats.csv \;
rm -rf gpurun_out/r05c/prof
tail -15 gpurun_out/r05c/eval_tests.log; cat gpurun_out/r05c/bench_stgcn.json; head -12 gpurun_out/r05c/stgcn_kernel_stats.csv | cut -c1-180
