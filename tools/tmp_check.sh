mkdir -p gpurun_out
v() { python -c "import sys, json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
{
echo "== fused_step_boundary test"
python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "fused_step_boundary" 2>&1 | grep -E "fused step boundary\]|passed|failed"
echo "== bench cfg2: multi-step | per-step | layers=0, 2 rounds"
for r in 1 2; do
echo "multi-step: $(python bench.py --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)"
echo "per-step:   $(REGENNET_LAYERS_STEPS=0 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)"
echo "layers=0:   $(REGENNET_LAYERS=0 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)"
done
for B in 16 32 48 64 96 128; do
echo "B=$B 250 steps: multi $(python bench.py --batch $B --respacing 250 --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v) | per-step $(REGENNET_LAYERS_STEPS=0 python bench.py --batch $B --respacing 250 --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v) | layers=0 $(REGENNET_LAYERS=0 python bench.py --batch $B --respacing 250 --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)"
done
} > gpurun_out/tmp_check.txt 2>&1
cat gpurun_out/tmp_check.txt
