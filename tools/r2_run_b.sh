# round 2, GPU run B: parity with the row-complete bulk kernels, bench A/B, kernel stats
set -u
R=$PWD; O=$R/gpurun_out/r2b; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|\[loop err\]|\[x3-tail|\[bench shape" $O/pytest.log | tail -60
REGENNET_BULK_RESID_LO=0 timeout 600 python -m pytest tests -m gpu -x -q -s -k "switch_point_sweep" > $O/pytest_residhi.log 2>&1; echo "pytest residhi rc=$?"
grep -E "passed|failed|\[x3-tail" $O/pytest_residhi.log | tail
timeout 600 python bench.py --no-cpu-baseline > $O/bench_rowgemm.json 2> $O/bench_rowgemm.err; echo "bench rc=$?"; cat $O/bench_rowgemm.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print(' ',e) for e in d['roofline']['per_kernel']]"
REGENNET_NO_ROWGEMM=1 timeout 300 python bench.py --no-cpu-baseline --profile-evals 0 2>/dev/null | head -c 200; echo
REGENNET_BULK_RESID_LO=0 timeout 300 python bench.py --no-cpu-baseline --profile-evals 0 2>/dev/null | head -c 200; echo
for n in 1 2 3 6; do REGENNET_STREAMS=$n timeout 300 python bench.py --no-cpu-baseline --profile-evals 0 --steps 1 2>/dev/null | head -c 120; echo " streams=$n"; done
cd /tmp && export TMPDIR=/tmp
for n in 1 4; do
  REGENNET_STREAMS=$n timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bulk_s$n -- python $R/bench.py --respacing 50 --x3-tail 0 --steps 2 --warmup 1 --no-cpu-baseline --profile-evals 0 > $O/bulk_s$n.log 2>&1 < /dev/null; echo "prof$n rc=$?"
done
rm -f $O/*kernel_trace.csv
head -8 $O/bulk_s1_kernel_stats.csv | cut -c1-160
