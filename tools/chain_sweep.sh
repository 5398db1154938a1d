# Chain-count sweep (REGENNET_STREAMS) of the headline workload + raw rocprofv3 kernel traces for 1 and 4 chains (note: the
# profiler serialises kernels, so its timestamps do not show chain concurrency).
set -u
R=$PWD; O=$R/gpurun_out/gaps; mkdir -p $O
for s in 1 2 3 4 6 8; do REGENNET_STREAMS=$s python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-evals 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams', $s, d['value'], d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
for s in 1 4; do REGENNET_STREAMS=$s timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o s$s -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --profile-evals 0 --respacing 50 --x3-tail 0 > $O/s$s.log 2>&1 < /dev/null; echo rc=$?; done
ls -la $O
