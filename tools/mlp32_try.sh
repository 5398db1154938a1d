# one GPU call: correctness of both layer-tail kernels against the fp64 host evaluation, alternating timings, stamps of the 32-row build
set -u
mkdir -p gpurun_out
{
echo "== check (M = 15360, 3 x 64 rows vs fp64 host)"
REGENNET_MLP_ROWS=64 timeout 120 tools/bin/mlp_bench 15360 20 1
REGENNET_MLP_ROWS=32 timeout 120 tools/bin/mlp_bench 15360 20 1
REGENNET_MLP_ROWS=32 timeout 120 tools/bin/mlp_bench 15330 20 1
echo "== alternating timings, 50 launches each"
for i in 1 2 3; do
  REGENNET_MLP_ROWS=64 timeout 120 tools/bin/mlp_bench 15360 50
  REGENNET_MLP_ROWS=32 timeout 120 tools/bin/mlp_bench 15360 50
done
echo "== M = 7680 (one of two chains)"
REGENNET_MLP_ROWS=64 timeout 120 tools/bin/mlp_bench 7680 50
REGENNET_MLP_ROWS=32 timeout 120 tools/bin/mlp_bench 7680 50
echo "== stamps (32-row)"
REGENNET_MLP_ROWS=32 timeout 120 tools/bin/mlp_bench_stamps 15360 20
} > gpurun_out/mlp32_try.txt 2>&1
cat gpurun_out/mlp32_try.txt
