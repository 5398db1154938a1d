for B in 13 16 24 32 64; do
for R in 768 8192; do
echo -n "B=$B SB_ROWS=$R: "
REGENNET_SB_ROWS=$R python bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --profile-evals 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
