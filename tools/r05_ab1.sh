# round 5, GPU call 2: same-box A/B of the k_layers variants (Wout / Wx L2 prefetch, static priority for the younger half) + kernel stats of the recogniser
mkdir -p gpurun_out/r05b
export TMPDIR=/tmp
bash tools/ab_many.sh 3 build/lib_base.so build/lib_pf.so build/lib_prio.so build/lib_pfprio.so > gpurun_out/r05b/ab_layers.txt 2>&1
bash tools/ab_many.sh 2 build/lib_base.so build/lib_pf.so build/lib_prio.so build/lib_pfprio.so -- --config ntu_action --guided --sampler ddim --respacing ddim100 > gpurun_out/r05b/ab_layers_cfg3.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05b/prof_stgcn -o stgcn -- python $GRAFT_REPO_ROOT/bench.py --config stgcn --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r05b/stgcn_prof.log 2>&1)
find gpurun_out/r05b/prof_stgcn -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05b/stgcn_kernel_stats.csv \;
rm -rf gpurun_out/r05b/prof_stgcn
cat gpurun_out/r05b/ab_layers.txt
