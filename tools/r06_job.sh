mkdir -p gpurun_out/r06
O=gpurun_out/r06/f16_speed.txt; : > $O
run() { python bench.py --no-cpu-baseline --profile-evals 0 --no-row-check "$@" 2>/dev/null | python -c "import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for r in 1 2 3; do
  for f in 0 1; do echo "cfg2 tail5 f16=$f $(run --steps 2 --warmup 1 --bulk-f16 $f)" >> $O; done
done
for f in 0 1; do for t in 5 2 1; do echo "cfg2 f16=$f tail=$t $(run --steps 2 --warmup 1 --bulk-f16 $f --x3-tail $t)" >> $O; done; done
for f in 0 1; do for t in 5 2 1; do echo "cfg3 f16=$f tail=$t $(run --steps 3 --warmup 1 --bulk-f16 $f --x3-tail $t --config ntu_action --sampler ddim --respacing ddim100 --guided)" >> $O; done; done
for f in 0 1; do for t in 3 2 1; do echo "eval_ddim5 f16=$f tail=$t $(run --steps 20 --warmup 3 --bulk-f16 $f --x3-tail $t --respacing ddim5)" >> $O; done; done
cat $O
bash tools/ab_trees.sh 3 build/r4_tree . -- --config chi3d --batch 128 --steps 2 --warmup 1 > gpurun_out/r06/cfg4_r4_vs_head.txt 2>&1; cat gpurun_out/r06/cfg4_r4_vs_head.txt
