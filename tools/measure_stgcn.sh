# The recogniser's measurement set (run on the GPU box): bench line, evaluation-batch line, rocprofv3 kernel stats, PMC summary -> gpurun_out/stgcn_final/
set -u
R=$PWD; O=$R/gpurun_out/stgcn_final
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python bench.py --config stgcn --steps 10 --warmup 2 > $O/bench_stgcn.json 2>/dev/null; head -c 220 $O/bench_stgcn.json; echo
python bench.py --config eval_pipeline --steps 10 --warmup 2 > $O/bench_eval_pipeline.json 2>/dev/null; head -c 300 $O/bench_eval_pipeline.json; echo
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o stgcn -- python $R/bench.py --config stgcn --steps 3 --warmup 1 > $O/stgcn_prof.log 2>&1)
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/stgcn_kernel_stats.csv \;
rm -rf $O/prof
bash tools/pmc_stgcn.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc_stgcn/summary.json $O/stgcn_pmc.json; tail -12 $O/pmc.log | cut -c1-260
