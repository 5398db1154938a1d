set -u
mkdir -p gpurun_out
{
echo "== check vs fp64 host (kernel 2, hres variant 0 | 1; kernel 3)"
REGENNET_MLP_KERNEL=2 timeout 120 tools/bin/mlp_bench_h0 15330 20 1
REGENNET_MLP_KERNEL=2 timeout 120 tools/bin/mlp_bench_h1 15330 20 1
REGENNET_MLP_KERNEL=3 timeout 120 tools/bin/mlp_bench_h0 15330 20 1
echo "== alternating timings, 50 launches each (kernel 1 | 2 h0 | 2 h1)"
for i in 1 2 3; do
  REGENNET_MLP_KERNEL=1 timeout 120 tools/bin/mlp_bench_h0 15360 50
  REGENNET_MLP_KERNEL=2 timeout 120 tools/bin/mlp_bench_h0 15360 50
  REGENNET_MLP_KERNEL=2 timeout 120 tools/bin/mlp_bench_h1 15360 50
done
echo "== stamps (kernel 2, h0 then h1)"
REGENNET_MLP_KERNEL=2 timeout 120 tools/bin/mlp_bench_h0_stamps 15360 20
REGENNET_MLP_KERNEL=2 timeout 120 tools/bin/mlp_bench_h1_stamps 15360 20
echo "== one workgroup alone (M = 64), stamps"
REGENNET_MLP_KERNEL=2 timeout 120 tools/bin/mlp_bench_h0_stamps 64 20
} > gpurun_out/mlp2_try2.txt 2>&1
cat gpurun_out/mlp2_try2.txt
