mkdir -p gpurun_out
v() { python -c "import sys, json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
{
python -m pytest tests/test_layers_gpu.py -m gpu -x -q -s 2>&1 | grep -E "k_layers|passed|failed|Error|error" | head -40
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "bench_shape or fused_step or row_independent" 2>&1 | tail -3
for r in 1 2; do
echo "cfg3 guided in-launch: $(python bench.py --config ntu_action --sampler ddim --respacing ddim100 --guided --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)"
echo "cfg3 per-step:         $(REGENNET_LAYERS_GUIDED=0 python bench.py --config ntu_action --sampler ddim --respacing ddim100 --guided --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)"
done
echo "cfg2: $(python bench.py --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)"
} > gpurun_out/tmp_check.txt 2>&1
cat gpurun_out/tmp_check.txt
