# rocprofv3 kernel stats of the uniform split-bf16 bench (20 steps) with and without k_mlp_x3, then the A/B lines
set -u
mkdir -p gpurun_out/x3prof
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in 1 0; do
REGENNET_MLP_X3=$e REGENNET_STREAMS=${STREAMS:-} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/x3prof -o e$e -- python $R/bench.py --precision bf16x3 --respacing 20 --no-cpu-baseline --profile-evals 0 --steps 1 --warmup 1 > $R/gpurun_out/x3prof/e$e.log 2>&1 < /dev/null
f=$(ls $R/gpurun_out/x3prof/*e${e}_kernel_stats.csv 2>/dev/null | head -1)
echo "== MLP_X3=$e  $f"
python - "$f" <<'P'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']:>6s}%")
P
done
cd $R
v() { python -c "import sys, json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for r in 1 2; do for e in 0 1; do
echo "MLP_X3=$e uniform-x3 20-step: $(REGENNET_MLP_X3=$e python bench.py --precision bf16x3 --respacing 20 --no-cpu-baseline --profile-evals 0 --steps 3 --warmup 1 2>/dev/null | v)"
done; done
