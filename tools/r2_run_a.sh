# round 2, GPU run A: parity suite, L2-path microbench, bench (schedule vs uniform x3), kernel stats of the bulk phase
set -u
R=$PWD; O=$R/gpurun_out/r2a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 > $O/pytest.log; echo "pytest rc=$?"
tail -5 $O/pytest.log
timeout 300 tools/bin/l2_paths_bench > $O/l2_paths.log 2>&1; echo "l2 rc=$?"
timeout 600 python bench.py > $O/bench_x3tail.json 2> $O/bench_x3tail.err; echo "bench rc=$?"; tail -c 1500 $O/bench_x3tail.json
timeout 300 python bench.py --precision bf16x3 --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.err; echo "bench x3 rc=$?"; head -c 400 $O/bench_x3.json
REGENNET_BULK_RESID_LO=0 timeout 300 python bench.py --no-cpu-baseline > $O/bench_x3tail_residhi.json 2>/dev/null; head -c 300 $O/bench_x3tail_residhi.json
for t in 0 1000; do
  timeout 300 python bench.py --x3-tail $t --no-cpu-baseline --steps 1 --profile-evals 0 2>/dev/null | head -c 250; echo
done
BF16=1 timeout 120 tools/bin/gemm_bench 0 > $O/gemm_bf16_v0.log 2>&1
BF16=1 timeout 120 tools/bin/gemm_bench 1 > $O/gemm_bf16_v1.log 2>&1
BF16=1 PLANES=1 PLANES_ONLY=1 timeout 120 tools/bin/gemm_bench 0 > $O/gemm_bf16_v0_planes.log 2>&1
timeout 120 tools/bin/gemm_bench 0 > $O/gemm_x3_v0.log 2>&1
cd /tmp && export TMPDIR=/tmp
for n in 1 4; do
  REGENNET_STREAMS=$n timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bulk_s$n -- python $R/bench.py --respacing 50 --x3-tail 0 --steps 2 --warmup 1 --no-cpu-baseline --profile-evals 0 > $O/bulk_s$n.log 2>&1 < /dev/null; echo "prof$n rc=$?"
done
rm -f $O/*kernel_trace.csv $O/*/*kernel_trace.csv
ls $O
