# round 5, GPU call: parity of the k_layers forms with the shadow RNG + LDS-DMA prefetch, then same-box A/B (base | prefetch | prefetch + shadow RNG)
mkdir -p gpurun_out/r05e
export TMPDIR=/tmp
python -m pytest tests/test_layers_gpu.py -x -q -s > gpurun_out/r05e/layers_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05e/layers_tests.log
tail -4 gpurun_out/r05e/layers_tests.log
python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r05e/bench_cfg2.json 2> gpurun_out/r05e/bench_cfg2.err; head -c 400 gpurun_out/r05e/bench_cfg2.json; echo
bash tools/ab_many.sh 3 build/lib_base.so build/lib_pf.so build/lib_new.so > gpurun_out/r05e/ab_shadow.txt 2>&1
bash tools/ab_many.sh 2 build/lib_base.so build/lib_pf.so build/lib_new.so -- --config ntu_action --guided --sampler ddim --respacing ddim100 > gpurun_out/r05e/ab_shadow_cfg3.txt 2>&1
cat gpurun_out/r05e/ab_shadow.txt gpurun_out/r05e/ab_shadow_cfg3.txt
