# k_qkv_attn_rx3 (REGENNET_QKV_RX3=1): parity subset through the throughput engine, then bench A/B
set -u
mkdir -p gpurun_out
REGENNET_QKV_RX3=2 timeout 600 python -m pytest tests/test_hip_parity.py -q -x -k "throughput and (ntu or headline or dispatch or sequence_length) and not switch_point and not chi3d and not text150" 2>&1 | tail -8 > gpurun_out/rx3_tests.txt
cat gpurun_out/rx3_tests.txt
v() { python -c "import sys, json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for r in 1 2; do for e in 0 2; do
export REGENNET_QKV_RX3=$e
echo "RX3=$e uniform-x3 20-step: $(python bench.py --precision bf16x3 --respacing 20 --no-cpu-baseline --profile-evals 0 --steps 3 --warmup 1 2>/dev/null | v)"
echo "RX3=$e eval ddim5:         $(python bench.py --respacing ddim5 --no-cpu-baseline --profile-evals 0 --steps 20 --warmup 3 2>/dev/null | v)"
echo "RX3=$e cfg3:               $(python bench.py --config ntu_action --sampler ddim --respacing ddim100 --guided --no-cpu-baseline --profile-evals 0 --steps 3 --warmup 1 2>/dev/null | v)"
done; done
