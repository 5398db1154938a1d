set -u
mkdir -p gpurun_out
v() { python -c "import sys, json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
{
echo "== fused decoder stack vs the kernel-per-stage chain"
timeout 120 tools/bin/layers_bench 256 60 1 20
timeout 120 tools/bin/layers_bench 256 60 8 20
timeout 120 tools/bin/layers_bench 7 37 2 5
timeout 120 tools/bin/layers_bench 128 64 8 20
timeout 120 tools/bin/layers_bench 512 60 8 10
echo "== stamps"
timeout 120 tools/bin/layers_bench_stamps 256 60 8 5
echo "== bench.py cfg2, REGENNET_LAYERS 0 | 1, 2 rounds"
for r in 1 2; do for k in 0 1; do
  echo "layers=$k: $(REGENNET_LAYERS=$k timeout 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --profile-evals 0 2>/dev/null | v)"
done; done
echo "== parity tests with REGENNET_LAYERS=1"
REGENNET_LAYERS=1 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "ntu or tiny or bench_shape" 2>&1 | tail -8
} > gpurun_out/layers_try.txt 2>&1
cat gpurun_out/layers_try.txt
