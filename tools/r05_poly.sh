# round 5: polyphase output written by k_sg_tconv's epilogue (tail 3) against fp32 convolution + k_sg_post for the two blocks in front of a stride-2 block, same box
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_eval_gpu.py -x -q 2>&1 | tail -2
for rep in 1 2; do
  for v in 0 1; do REGENNET_SG_NO_POLY_TAIL=$v timeout 300 python bench.py --config stgcn --steps 10 --warmup 2 2>/dev/null | python -c "import sys, json; print('  NO_POLY_TAIL=$v ms/forward', json.loads(sys.stdin.readline())['ms_per_step'])"; done
done
