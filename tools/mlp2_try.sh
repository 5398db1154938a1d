set -u
mkdir -p gpurun_out
v() { python -c "import sys, json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
{
echo "== check vs fp64 host (kernel 1 = rgn_mlp.hip, 2 = mlp2 64-row, 3 = mlp2 32-row)"
for k in 1 2 3; do REGENNET_MLP_KERNEL=$k timeout 120 tools/bin/mlp_bench 15360 20 1; done
REGENNET_MLP_KERNEL=2 timeout 120 tools/bin/mlp_bench 15330 20 1
REGENNET_MLP_KERNEL=2 timeout 120 tools/bin/mlp_bench 1000 20 1
echo "== alternating timings, 50 launches each (1 | 2 | 3)"
for i in 1 2 3; do
  for k in 1 2 3; do REGENNET_MLP_KERNEL=$k timeout 120 tools/bin/mlp_bench 15360 50; done
done
echo "== M = 7680"
for k in 1 2 3; do REGENNET_MLP_KERNEL=$k timeout 120 tools/bin/mlp_bench 7680 50; done
echo "== stamps (kernel 2)"
REGENNET_MLP_KERNEL=2 timeout 120 tools/bin/mlp_bench_stamps 15360 20
echo "== bench.py cfg2, kernel 1 | 2, 3 rounds"
for r in 1 2 3; do for k in 1 2; do
  echo "kernel=$k: $(REGENNET_MLP_KERNEL=$k python bench.py --no-cpu-baseline --steps 2 --warmup 1 --profile-evals 0 2>/dev/null | v)"
done; done
} > gpurun_out/mlp2_try.txt 2>&1
cat gpurun_out/mlp2_try.txt
