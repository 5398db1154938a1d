# full GPU suite + smoke + the quick bench lines, one call
set -u
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/gputest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1
v() { python -c "import sys, json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('kernel','')[:16], (d.get('roofline') or {}).get('frac'))"; }
{
echo "cfg2: $(python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | v)"
echo "cfg3: $(python bench.py --config ntu_action --sampler ddim --respacing ddim100 --guided --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | v)"
} > gpurun_out/bench_lines.txt 2>&1
cat gpurun_out/gputest.txt gpurun_out/smoke.txt gpurun_out/bench_lines.txt
