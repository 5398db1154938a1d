mkdir -p gpurun_out/r06
python -m pytest tests/test_precision_gpu.py tests/test_hip_parity.py tests/test_layers_gpu.py -q -x -s -k "kernel_per_stage_chain or evaluation_setting_switch or fused_step_boundary or one_and_two_sample or decoder_stack" 2>&1 | grep -v amdgpu.ids | grep "^\[\|passed\|failed\|Error\|assert" | tail -30
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['dtype'][-60:])"; }
for B in 16 32 48; do for f in "" "--f16-steps 0"; do python bench.py --batch $B --respacing ddim5 --no-cpu-baseline --steps 20 --warmup 3 --profile-evals 0 --no-row-check $f 2>/dev/null | p "eval_ddim5 B=$B $f" | tee -a gpurun_out/r06/eval_small_batches.txt; done; done
