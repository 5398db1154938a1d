set -u
R=$PWD; O=$R/gpurun_out/r2d; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for b in 1 4 16 64; do timeout 300 python bench.py --batch $b --no-cpu-baseline --profile-evals 0 --steps 1 2>/dev/null | head -c 130; echo " B=$b"; done
timeout 300 python bench.py --config ntu_action --sampler ddim --respacing ddim100 --guided --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | head -c 200; echo " cfg3"
timeout 300 python bench.py --config chi3d --batch 128 --no-cpu-baseline --steps 1 --warmup 1 --profile-evals 0 2>/dev/null | head -c 200; echo " cfg4"
timeout 300 python bench.py --config text150 --batch 256 --sampler ddim --respacing ddim50 --guided --no-cpu-baseline --steps 3 --warmup 1 --profile-evals 0 2>/dev/null | head -c 200; echo " cfg5"
