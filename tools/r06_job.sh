mkdir -p gpurun_out/r06
python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/r06/gpu_suite_full2.txt; tail -8 gpurun_out/r06/gpu_suite_full2.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee gpurun_out/r06/smoke.txt
