set -u
v() { python -c "import sys, json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for r in 1 2; do for s in 1 2 4; do
export REGENNET_STREAMS=$s
echo "chains=$s uniform-x3 20-step: $(python bench.py --precision bf16x3 --respacing 20 --no-cpu-baseline --profile-evals 0 --steps 3 --warmup 1 2>/dev/null | v)"
echo "chains=$s eval ddim5:         $(python bench.py --respacing ddim5 --no-cpu-baseline --profile-evals 0 --steps 20 --warmup 3 2>/dev/null | v)"
echo "chains=$s cfg3:               $(python bench.py --config ntu_action --sampler ddim --respacing ddim100 --guided --no-cpu-baseline --profile-evals 0 --steps 3 --warmup 1 2>/dev/null | v)"
done; done
unset REGENNET_STREAMS
echo "default       eval ddim5:   $(python bench.py --respacing ddim5 --no-cpu-baseline --profile-evals 0 --steps 20 --warmup 3 2>/dev/null | v)"
