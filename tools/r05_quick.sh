# parity tests of the recogniser + two bench lines (+ optional stamps of the -DRGN_SG_PROF build if build/lib_sgprof.so exists)
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_eval_gpu.py -x -q 2>&1 | tail -2
for rep in 1 2; do timeout 300 python bench.py --config stgcn --steps 10 --warmup 2 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('  ms/forward', d['ms_per_step'], [p['ms_per_forward'] for p in d['per_length']])"; done
if [ -f build/lib_sgprof.so ]; then bash tools/r05_tconv_stamps.sh | tail -4; fi
