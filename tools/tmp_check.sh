mkdir -p gpurun_out
v() { python -c "import sys, json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
{
python -m pytest tests/test_hip_parity.py -m gpu -x -q -s -k "reference_evaluation_setting" 2>&1 | grep -E "5-step|passed|failed|Error|assert" | head -20
echo "== bench NTU B=256, ddim5 schedule through p_sample_loop: tail 5 (default) | 3 | 2 | 0"
for t in 5 3 2 0; do echo "tail $t: $(python bench.py --respacing ddim5 --x3-tail $t --no-cpu-baseline --steps 20 --warmup 3 --profile-evals 0 2>/dev/null | v)"; done
} > gpurun_out/tmp_check.txt 2>&1
cat gpurun_out/tmp_check.txt
