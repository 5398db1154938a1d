mkdir -p gpurun_out
{
echo "== k_qkv_attn_rs output hash (round 3 recorded c0ab23edee7c619b at 256, be02817a69a0f51b at 128)"
RS=1 BF16=1 timeout 100 tools/bin/qkv_attn_bench 256 60 50
RS=1 BF16=1 timeout 100 tools/bin/qkv_attn_bench 128 60 50
python -m pytest tests/test_hip_parity.py -m gpu -x -q -s -k "random_8_layer" 2>&1 | grep -E "default schedule|passed|failed|rror" | head
} > gpurun_out/tmp_check.txt 2>&1
cat gpurun_out/tmp_check.txt
