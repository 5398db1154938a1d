set -u
mkdir -p gpurun_out
v() { python -c "import sys, json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
{
echo "== fused decoder stack vs the kernel-per-stage chain"
timeout 120 tools/bin/layers_bench 256 60 8 20
timeout 120 tools/bin/layers_bench 7 37 2 5
echo "== stamps"
timeout 120 tools/bin/layers_bench_stamps 256 60 8 5
echo "== bench.py cfg2, 2 rounds"
for r in 1 2; do
  echo "multi-step: $(timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)"
done
echo "== parity tests (fused path forced for small batches)"
REGENNET_LAYERS_MIN_B=1 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "(ntu and loop) or bench_shape or fused_step" 2>&1 | tail -4
} > gpurun_out/layers_try.txt 2>&1
cat gpurun_out/layers_try.txt
