# A round's measurement set (run on the GPU box; tools/measure_round.sh TAG, e.g. r06): bench lines of the BASELINE configs + the evaluation setting + the recogniser, rocprofv3 kernel
# stats of the default bench command and of the other configs, PMC passes (cfg2: counted on the FULL-LENGTH launch of the plain-bf16 phase). Outputs under
# gpurun_out/final_TAG/; what should be judged is copied into profiles/TAG_*.
set -u
TAG=${1:-r06}
R=$PWD; O=$R/gpurun_out/final_$TAG
rm -rf $O; mkdir -p $O
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err < /dev/null; echo bench rc=$?
head -c 300 $O/bench_cfg2.json; echo
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg2_driver_style.json 2>/dev/null; head -c 160 $O/bench_cfg2_driver_style.json; echo
python bench.py --config ntu_action --sampler ddim --respacing ddim100 --guided --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_cfg3.json 2>/dev/null; head -c 160 $O/bench_cfg3.json; echo
python bench.py --config chi3d --batch 128 --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_cfg4.json 2>/dev/null; head -c 160 $O/bench_cfg4.json; echo
python bench.py --config text150 --batch 256 --sampler ddim --respacing ddim50 --guided --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_cfg5.json 2>/dev/null; head -c 160 $O/bench_cfg5.json; echo
python bench.py --batch 1 --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_cfg1_B1.json 2>/dev/null; head -c 160 $O/bench_cfg1_B1.json; echo
python bench.py --respacing ddim5 --no-cpu-baseline --steps 20 --warmup 3 --profile-evals 0 > $O/bench_eval_ddim5.json 2>/dev/null; head -c 160 $O/bench_eval_ddim5.json; echo
python bench.py --config stgcn --steps 10 --warmup 2 > $O/bench_stgcn.json 2>/dev/null; head -c 200 $O/bench_stgcn.json; echo
python bench.py --config eval_pipeline --steps 10 --warmup 2 > $O/bench_eval_pipeline.json 2>/dev/null; head -c 200 $O/bench_eval_pipeline.json; echo
python bench.py --config stgcn --recogniser-f16 --steps 10 --warmup 2 > $O/bench_stgcn_fp16_form.json 2>/dev/null; head -c 200 $O/bench_stgcn_fp16_form.json; echo
python bench.py --config eval_pipeline --recogniser-f16 --steps 10 --warmup 2 > $O/bench_eval_pipeline_fp16_recogniser.json 2>/dev/null; head -c 200 $O/bench_eval_pipeline_fp16_recogniser.json; echo
bash tools/batch_sweep.sh > $O/batch_sweep.txt 2>&1; cat $O/batch_sweep.txt
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench flags...
  n=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $n -- python $R/bench.py --no-cpu-baseline --no-row-check "$@" > $O/$n.log 2>&1 < /dev/null; echo "$n rc=$?"
}
prof cfg2_default
prof cfg3 --steps 1 --warmup 1 --profile-evals 0 --config ntu_action --sampler ddim --respacing ddim100 --guided
prof cfg4 --steps 1 --warmup 1 --profile-evals 0 --config chi3d --batch 128 --respacing 50
prof cfg5 --steps 1 --warmup 1 --profile-evals 0 --config text150 --batch 256 --sampler ddim --respacing ddim50 --guided
prof eval_ddim5 --steps 10 --warmup 2 --profile-evals 0 --respacing ddim5
prof stgcn --config stgcn --steps 3 --warmup 1
prof stgcn_fp16_form --config stgcn --recogniser-f16 --steps 3 --warmup 1
rm -f $O/*kernel_trace.csv $O/*agent_info.csv $O/*domain_stats.csv
cd $R
PMC_FULL=1 PMC_STEPS=990 bash tools/collect_pmc.sh gpurun_out/final_$TAG/pmc_bench.json ntu_B256_bf16_x3tail_plain 2>&1 | grep "rc="
mkdir -p $O/pmc_raw && cp gpurun_out/pmc/*counter_collection.csv $O/pmc_raw/ 2>/dev/null
ls $O
