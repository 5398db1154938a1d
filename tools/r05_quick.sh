# parity tests of the recogniser + A/B of one environment switch: tools/r05_quick.sh REGENNET_<KEY>  (bench lines with the switch unset / set to 1, two rounds)
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_eval_gpu.py -x -q 2>&1 | tail -2
for rep in 1 2; do
  for v in "" 1; do
    if [ -n "$v" ]; then export $1=$v; else unset $1; fi
    timeout 300 python bench.py --config stgcn --steps 10 --warmup 2 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('  $1=$v ms/forward', [p['ms_per_forward'] for p in d['per_length']])"
  done
done
