# same-box A/B of the 64-row and the 32-row layer tail under bench.py (env switch, one library), plus an M sweep of the kernels alone
set -u
mkdir -p gpurun_out
v() { python -c "import sys, json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
{
echo "== M sweep (us per launch: 64-row | 32-row)"
for M in 1920 3840 5760 7680 9600 11520 13440 15360 19200 30720; do
  a=$(REGENNET_MLP_ROWS=64 timeout 120 tools/bin/mlp_bench $M 50 | awk '{print $3}')
  b=$(REGENNET_MLP_ROWS=32 timeout 120 tools/bin/mlp_bench $M 50 | awk '{print $3}')
  echo "M=$M  $a | $b"
done
echo "== bench.py cfg2 (B=256, 1000 steps), rows x streams, 2 rounds"
for r in 1 2; do for rows in 64 32; do for st in 1 2; do
  echo "rows=$rows streams=$st: $(REGENNET_MLP_ROWS=$rows REGENNET_STREAMS=$st python bench.py --no-cpu-baseline --steps 2 --warmup 1 --profile-evals 0 2>/dev/null | v)"
done; done; done
echo "== bench.py B=128 / 64 / 32, 250 steps, default streams"
for B in 128 64 32; do for rows in 64 32; do
  echo "B=$B rows=$rows: $(REGENNET_MLP_ROWS=$rows python bench.py --batch $B --respacing 250 --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | v)"
done; done
} > gpurun_out/mlp32_ab.txt 2>&1
cat gpurun_out/mlp32_ab.txt
