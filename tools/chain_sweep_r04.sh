mkdir -p gpurun_out
v() { python -c "import sys, json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
{
for st in 2 3 4 6; do
echo "cfg4 streams=$st: $(REGENNET_STREAMS=$st python bench.py --config chi3d --batch 128 --respacing 250 --no-cpu-baseline --steps 2 --warmup 1 --profile-evals 0 2>/dev/null | v)"
done
for st in 2 3 4 6; do
echo "cfg5 streams=$st: $(REGENNET_STREAMS=$st python bench.py --config text150 --batch 256 --sampler ddim --respacing ddim50 --guided --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)"
done
echo "cfg4 default: $(python bench.py --config chi3d --batch 128 --respacing 250 --no-cpu-baseline --steps 2 --warmup 1 --profile-evals 0 2>/dev/null | v)"
echo "cfg5 default: $(python bench.py --config text150 --batch 256 --sampler ddim --respacing ddim50 --guided --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)"
echo "cfg4 mlp32: $(REGENNET_MLP_ROWS=32 python bench.py --config chi3d --batch 128 --respacing 250 --no-cpu-baseline --steps 2 --warmup 1 --profile-evals 0 2>/dev/null | v)"
} > gpurun_out/chain_sweep_r04.txt 2>&1
cat gpurun_out/chain_sweep_r04.txt
