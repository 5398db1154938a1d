mkdir -p gpurun_out
{
for r in 1 2 3; do for M in 15360 19200; do
echo "M=$M base: $(timeout 100 tools/bin/mlp_bench $M 50 | head -1)   dma-first: $(timeout 100 tools/bin/mlp_bench_df $M 50 | head -1)"
done; done
timeout 100 tools/bin/mlp_bench_df_stamps 15360 20 | tail -3
} > gpurun_out/tmp_check.txt 2>&1
cat gpurun_out/tmp_check.txt
