mkdir -p gpurun_out/r06
python -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/r06/gpu_suite.txt; tail -5 gpurun_out/r06/gpu_suite.txt
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['unit'], d['ms_per_step'], d['dtype'], d.get('headline_row_check_max_abs'))"; }
python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | p cfg2 | tee gpurun_out/r06/bench_quick.txt
python bench.py --no-cpu-baseline --steps 3 --warmup 1 --f16-steps 0 2>/dev/null | p cfg2_bf16rule | tee -a gpurun_out/r06/bench_quick.txt
python bench.py --config ntu_action --sampler ddim --respacing ddim100 --guided --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | p cfg3 | tee -a gpurun_out/r06/bench_quick.txt
python bench.py --config ntu_action --sampler ddim --respacing ddim100 --guided --no-cpu-baseline --steps 3 --warmup 1 --f16-steps 0 2>/dev/null | p cfg3_bf16rule | tee -a gpurun_out/r06/bench_quick.txt
python bench.py --respacing ddim5 --no-cpu-baseline --steps 20 --warmup 3 --profile-evals 0 2>/dev/null | p eval_ddim5 | tee -a gpurun_out/r06/bench_quick.txt
python bench.py --respacing ddim5 --no-cpu-baseline --steps 20 --warmup 3 --profile-evals 0 --f16-steps 0 2>/dev/null | p eval_ddim5_bf16rule | tee -a gpurun_out/r06/bench_quick.txt
