set -u
R=$PWD; O=$R/gpurun_out/r2c; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -k "sampling_loop or forward or bench_shape or row_independent or sequence_length" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print(' ',e) for e in d['roofline']['per_kernel']]"
