# Round-4 closing measurement set after k_mlp_x3 (run on the GPU box): bench lines of the BASELINE configs + the evaluation schedule + the uniform
# split-bf16 mode, rocprofv3 kernel stats of the default bench command / cfg3 / cfg5 / the tail phase / the uniform mode, PMC passes of the uniform
# split-bf16 workload (k_mlp_x3, k_qkv_attn<true>). Outputs under gpurun_out/final/; what should be judged is copied into profiles/r04_*.
set -u
R=$PWD; O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err < /dev/null; echo bench rc=$?
head -c 300 $O/bench_cfg2.json; echo
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_cfg2_driver_style.json 2>/dev/null; head -c 160 $O/bench_cfg2_driver_style.json; echo
python bench.py --config ntu_action --sampler ddim --respacing ddim100 --guided --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_cfg3.json 2>/dev/null; head -c 160 $O/bench_cfg3.json; echo
python bench.py --config chi3d --batch 128 --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_cfg4.json 2>/dev/null; head -c 160 $O/bench_cfg4.json; echo
python bench.py --config text150 --batch 256 --sampler ddim --respacing ddim50 --guided --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_cfg5.json 2>/dev/null; head -c 160 $O/bench_cfg5.json; echo
python bench.py --batch 1 --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_cfg1_B1.json 2>/dev/null; head -c 160 $O/bench_cfg1_B1.json; echo
python bench.py --respacing ddim5 --no-cpu-baseline --steps 20 --warmup 3 --profile-evals 0 > $O/bench_eval_ddim5.json 2>/dev/null; head -c 160 $O/bench_eval_ddim5.json; echo
python bench.py --precision bf16x3 --respacing 100 --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_cfg2_uniform_x3_100steps.json 2>/dev/null; head -c 160 $O/bench_cfg2_uniform_x3_100steps.json; echo
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench flags...
  n=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $n -- python $R/bench.py --no-cpu-baseline "$@" > $O/$n.log 2>&1 < /dev/null; echo "$n rc=$?"
}
prof cfg2_default
prof cfg3 --steps 1 --warmup 1 --profile-evals 0 --config ntu_action --sampler ddim --respacing ddim100 --guided
prof cfg5 --steps 1 --warmup 1 --profile-evals 0 --config text150 --batch 256 --sampler ddim --respacing ddim50 --guided
prof cfg2_tail --steps 1 --warmup 1 --profile-evals 0 --respacing 50 --x3-tail 50
rm -f $O/*kernel_trace.csv $O/*agent_info.csv $O/*domain_stats.csv
cd $R
bash tools/collect_pmc.sh gpurun_out/final/pmc_x3.json ntu_B256_bf16x3_uniform --precision bf16x3 2>&1 | grep "rc="
ls $O
