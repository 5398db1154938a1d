# Round-end measurement set (run on the GPU box): default bench line, rocprofv3 kernel stats with 1 and 4 kernel chains,
# PMC passes. Outputs under gpurun_out/final/; copy what should be judged into profiles/.
set -u
R=$PWD
mkdir -p $R/gpurun_out/final
python bench.py > $R/gpurun_out/final/bench.json 2> $R/gpurun_out/final/bench.err < /dev/null; echo bench rc=$?
tail -c 600 $R/gpurun_out/final/bench.json
cd /tmp && export TMPDIR=/tmp
for n in 1 4; do
  REGENNET_STREAMS=$n timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final -o s$n -- python $R/bench.py --respacing 50 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/final/s$n.log 2>&1 < /dev/null; echo prof$n rc=$?
done
cd $R
bash tools/collect_pmc.sh gpurun_out/final/pmc_summary.json 2>&1 | grep "rc=" 
mkdir -p gpurun_out/final/pmc_raw && cp gpurun_out/pmc/*counter_collection.csv gpurun_out/final/pmc_raw/
rm -f gpurun_out/final/*kernel_trace.csv
ls gpurun_out/final
