p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['unit'], d['ms_per_step'], d['config'].get('workload','')[:90])"; }
python bench.py --config ntu_action --sampler ddim --respacing ddim100 --guided --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | p cfg3
python bench.py --config chi3d --batch 128 --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | p cfg4
python bench.py --config text150 --batch 256 --sampler ddim --respacing ddim50 --guided --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | p cfg5
python bench.py --precision f32 --no-cpu-baseline --steps 1 --warmup 1 --respacing 100 2>/dev/null | p f32_100steps
python bench.py --batch 1 --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | p B1
python bench.py --batch 4 --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | p B4
python bench.py --batch 1024 --no-cpu-baseline --steps 1 --warmup 1 --respacing 100 2>/dev/null | p B1024_100steps
python bench.py --batch 512 --no-cpu-baseline --steps 1 --warmup 1 --respacing 100 2>/dev/null | p B512_100steps
