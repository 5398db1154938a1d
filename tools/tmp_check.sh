mkdir -p gpurun_out
{
python -m pytest tests/test_layers_gpu.py -m gpu -x -q -s 2>&1 | grep -E "k_layers|passed|failed|Error|error" | head -40
echo "== bench JSON"
python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1
} > gpurun_out/tmp_check.txt 2>&1
cat gpurun_out/tmp_check.txt
