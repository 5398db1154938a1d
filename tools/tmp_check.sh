mkdir -p gpurun_out
v() { python -c "import sys, json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
{
timeout 200 tools/bin/layers_bench_stamps 256 60 8 5 | tail -3
python -m pytest tests/test_layers_gpu.py -m gpu -x -q -k "quad_shared or kernel_per_stage_chain" 2>&1 | tail -3
for r in 1 2; do echo "cfg2: $(python bench.py --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)"; done
} > gpurun_out/tmp_check.txt 2>&1
cat gpurun_out/tmp_check.txt
