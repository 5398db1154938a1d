# k_mlp_x3 (REGENNET_MLP_X3=1): parity subset through the throughput engine, then bench A/B in uniform split-bf16 mode and on the evaluation schedule
set -u
mkdir -p gpurun_out
REGENNET_MLP_X3=1 timeout 600 python -m pytest tests/test_hip_parity.py -q -x -k "throughput and (ntu or chi3d or text150 or headline or dispatch) and not switch_point" 2>&1 | tail -12 > gpurun_out/x3_tests.txt
cat gpurun_out/x3_tests.txt
v() { python -c "import sys, json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
{
for r in 1 2; do
for e in 0 1; do
echo "MLP_X3=$e uniform-x3 20-step: $(REGENNET_MLP_X3=$e python bench.py --precision bf16x3 --respacing 20 --no-cpu-baseline --profile-evals 0 --steps 3 --warmup 1 2>/dev/null | v)"
echo "MLP_X3=$e eval ddim5:         $(REGENNET_MLP_X3=$e python bench.py --respacing ddim5 --no-cpu-baseline --profile-evals 0 --steps 20 --warmup 3 2>/dev/null | v)"
done; done
} > gpurun_out/x3_bench.txt 2>&1
cat gpurun_out/x3_bench.txt
