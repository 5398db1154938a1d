# full GPU suite + the bench lines of the configs, one call
set -u
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/gputest.txt
v() { python -c "import sys, json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('kernel','')[:12], (d.get('roofline') or {}).get('frac'))"; }
{
echo "cfg2: $(python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | v)"
echo "cfg2 streams=1: $(REGENNET_STREAMS=1 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)"
echo "cfg2 layers=0: $(REGENNET_LAYERS=0 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)"
echo "cfg3: $(python bench.py --config ntu_action --sampler ddim --respacing ddim100 --guided --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | v)"
echo "cfg3 layers=0: $(REGENNET_LAYERS=0 python bench.py --config ntu_action --sampler ddim --respacing ddim100 --guided --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)"
echo "cfg3 streams=1: $(REGENNET_STREAMS=1 python bench.py --config ntu_action --sampler ddim --respacing ddim100 --guided --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)"
for B in 64 128 192 512; do
echo "B=$B 250 steps: $(python bench.py --batch $B --respacing 250 --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)   layers=0: $(REGENNET_LAYERS=0 python bench.py --batch $B --respacing 250 --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)"
done
} > gpurun_out/bench_lines.txt 2>&1
cat gpurun_out/gputest.txt gpurun_out/bench_lines.txt
