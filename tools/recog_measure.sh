set -u
R=$PWD; O=$R/gpurun_out/r06/recog; rm -rf $O; mkdir -p $O
python bench.py --config stgcn --steps 10 --warmup 2 > $O/bench_stgcn.json 2>/dev/null; head -c 200 $O/bench_stgcn.json; echo
python bench.py --config stgcn --recogniser-f16 --steps 10 --warmup 2 > $O/bench_stgcn_fp16_form.json 2>/dev/null; head -c 200 $O/bench_stgcn_fp16_form.json; echo
python bench.py --config eval_pipeline --steps 10 --warmup 2 > $O/bench_eval_pipeline.json 2>/dev/null; head -c 200 $O/bench_eval_pipeline.json; echo
python bench.py --config eval_pipeline --recogniser-f16 --steps 10 --warmup 2 > $O/bench_eval_pipeline_fp16_recogniser.json 2>/dev/null; head -c 200 $O/bench_eval_pipeline_fp16_recogniser.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stgcn -- python $R/bench.py --config stgcn --steps 3 --warmup 1 > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stgcn_fp16_form -- python $R/bench.py --config stgcn --recogniser-f16 --steps 3 --warmup 1 > $O/p2.log 2>&1
rm -f $O/*agent_info.csv $O/*domain_stats.csv
ls $O
