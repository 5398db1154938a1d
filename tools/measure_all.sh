# Round measurement set (run on the GPU box): default bench line, rocprofv3 kernel stats (1 and 4 kernel chains) of the
# headline workload and of BASELINE configs 3/4/5, PMC passes. Outputs under gpurun_out/final/; copy what should be judged
# into profiles/ (named per round).
set -u
R=$PWD; O=$R/gpurun_out/final
mkdir -p $O
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err < /dev/null; echo bench rc=$?
head -c 400 $O/bench_cfg2.json; echo
python bench.py --config ntu_action --sampler ddim --respacing ddim100 --guided --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_cfg3.json 2>/dev/null; head -c 200 $O/bench_cfg3.json; echo
python bench.py --config chi3d --batch 128 --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_cfg4.json 2>/dev/null; head -c 200 $O/bench_cfg4.json; echo
python bench.py --config text150 --batch 256 --sampler ddim --respacing ddim50 --guided --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_cfg5.json 2>/dev/null; head -c 200 $O/bench_cfg5.json; echo
python bench.py --batch 1 --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_cfg1_B1.json 2>/dev/null; head -c 200 $O/bench_cfg1_B1.json; echo
python bench.py --batch 10 --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_B10.json 2>/dev/null; head -c 200 $O/bench_B10.json; echo
python bench.py --precision bf16x3 --no-cpu-baseline --steps 1 --warmup 1 --profile-evals 0 > $O/bench_cfg2_uniform_x3.json 2>/dev/null; head -c 200 $O/bench_cfg2_uniform_x3.json; echo
cd /tmp && export TMPDIR=/tmp
prof() {  # name, streams, bench flags...
  n=$1; st=$2; shift 2
  REGENNET_STREAMS=$st timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $n -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --profile-evals 0 "$@" > $O/$n.log 2>&1 < /dev/null; echo "$n rc=$?"
}
prof cfg2_bulk_s1 1 --respacing 50 --x3-tail 0
prof cfg2_bulk_s4 4 --respacing 50 --x3-tail 0
prof cfg2_tail_s1 1 --respacing 50 --x3-tail 50
prof cfg1_B1 1 --batch 1 --respacing 50 --x3-tail 0
prof cfg3_s4 4 --config ntu_action --sampler ddim --respacing ddim100 --guided
prof cfg4_s4 4 --config chi3d --batch 128 --respacing 50
prof cfg4_s1 1 --config chi3d --batch 128 --respacing 50 --x3-tail 0
prof cfg5_s4 4 --config text150 --batch 256 --sampler ddim --respacing ddim50 --guided
rm -f $O/*kernel_trace.csv $O/*agent_info.csv $O/*domain_stats.csv
cd $R
bash tools/collect_pmc.sh gpurun_out/final/pmc_bench.json ntu_B256_bf16_x3tail_plain 2>&1 | grep "rc="
mkdir -p $O/pmc_raw && cp gpurun_out/pmc/*counter_collection.csv $O/pmc_raw/
ls $O
