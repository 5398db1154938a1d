# round 5, GPU call 4: per-phase floor of a persistent single-XCD kernel; the workgroup-count quantisation of the one-kernel form (B = 256 / 320 / 384 / 512)
mkdir -p gpurun_out/r05d
export TMPDIR=/tmp
timeout 120 tools/bin/xcd_phase_bench 2000 > gpurun_out/r05d/xcd_phase_bench.txt 2>&1; echo "rc=$?" >> gpurun_out/r05d/xcd_phase_bench.txt
for B in 256 320 384 512; do
  python bench.py --batch $B --respacing 250 --steps 2 --warmup 1 --no-cpu-baseline --no-row-check --profile-evals 0 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('B', $B, 'motions/s', d['value'], 'ms_per_call', d['ms_per_step'])" >> gpurun_out/r05d/layers_quantisation.txt
done
cat gpurun_out/r05d/xcd_phase_bench.txt gpurun_out/r05d/layers_quantisation.txt
