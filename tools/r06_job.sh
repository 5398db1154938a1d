mkdir -p gpurun_out/r06
python -m pytest tests/test_precision_gpu.py tests/test_hip_parity.py -q -x -s -k "three_phase or full_size_shard_against or launch_sequence_rows or rccl_one_rank or row_independent" 2>&1 | grep -v amdgpu.ids | grep "^\[\|passed\|failed\|Error\|assert" | tee gpurun_out/r06/tests_150.txt
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['unit'], d['ms_per_step'], d['dtype'], d.get('headline_row_check_max_abs'))"; }
for r in 1 2; do
python bench.py --config text150 --batch 256 --sampler ddim --respacing ddim50 --guided --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | p cfg5 | tee -a gpurun_out/r06/bench_150.txt
python bench.py --config text150 --batch 256 --sampler ddim --respacing ddim50 --guided --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 --f16-steps 0 2>/dev/null | p cfg5_bf16rule | tee -a gpurun_out/r06/bench_150.txt
done
python bench.py --config chi3d --batch 128 --no-cpu-baseline --steps 2 --warmup 1 --profile-evals 0 2>/dev/null | p cfg4 | tee -a gpurun_out/r06/bench_150.txt
python bench.py --config chi3d --batch 128 --no-cpu-baseline --steps 2 --warmup 1 --profile-evals 0 --f16-steps 0 2>/dev/null | p cfg4_bf16rule | tee -a gpurun_out/r06/bench_150.txt
