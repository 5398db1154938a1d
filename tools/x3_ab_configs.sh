# k_mlp_x3 on / off (REGENNET_MLP_X3) on the workloads with a split-bf16 tail: same box, alternating
set -u
v() { python -c "import sys, json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for r in 1 2; do for e in 0 1; do
export REGENNET_MLP_X3=$e
echo "MLP_X3=$e cfg2 : $(python bench.py --no-cpu-baseline --profile-evals 0 --steps 2 --warmup 1 2>/dev/null | v)"
echo "MLP_X3=$e cfg3 : $(python bench.py --config ntu_action --sampler ddim --respacing ddim100 --guided --no-cpu-baseline --profile-evals 0 --steps 3 --warmup 1 2>/dev/null | v)"
echo "MLP_X3=$e cfg4 : $(python bench.py --config chi3d --batch 128 --no-cpu-baseline --profile-evals 0 --steps 2 --warmup 1 2>/dev/null | v)"
echo "MLP_X3=$e cfg5 : $(python bench.py --config text150 --batch 256 --sampler ddim --respacing ddim50 --guided --no-cpu-baseline --profile-evals 0 --steps 3 --warmup 1 2>/dev/null | v)"
echo "MLP_X3=$e eval : $(python bench.py --respacing ddim5 --no-cpu-baseline --profile-evals 0 --steps 20 --warmup 3 2>/dev/null | v)"
done; done
