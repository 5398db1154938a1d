#!/bin/bash
# scratch: kernel durations of the small-batch engine at B = 1, fused in_proj + attention on / off
O=$PWD/gpurun_out/sba; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
for F in 1 0; do
  REGENNET_SB_FUSED_ATTN=$F timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o f$F -- python $R/bench.py --batch 1 --no-cpu-baseline --steps 1 --warmup 1 --profile-evals 0 --respacing 100 > $O/prof_f$F.log 2>&1 < /dev/null
  echo "fused=$F rc=$?"; head -8 $O/f${F}_kernel_stats.csv < /dev/null | cut -c1-150
done
rm -f $O/*kernel_trace.csv $O/*agent_info.csv $O/*domain_stats.csv $O/*.db
