# motions/s and ms per step over the batch size (60 frames, 250-step DDPM): tools/batch_sweep.sh [extra bench flags]
# (the execution forms: small-batch engine B <= 10, kernel per stage below LAYERS_MIN_B, one workgroup per sample above)
for B in 1 2 4 8 10 16 32 48 64 96 128 192 256; do
  python bench.py --batch $B --respacing 250 --no-cpu-baseline --profile-evals 0 --no-row-check --steps 3 --warmup 1 "$@" 2>/dev/null | \
    python -c "import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B', d['value'], 'motions/s', d['ms_per_step'], 'ms per call', round(d['ms_per_step']/250*1000, 1), 'us per step')"
done
