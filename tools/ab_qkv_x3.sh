# A/B of the split phase's in_proj + attention kernel on the evaluation schedule (bench.py --respacing ddim5): direct-to-LDS k_qkv_attn<true> (REGENNET_QKV_X3_DMA=1) |
# k_qkv_attn_rs_x3<2> (default) | <1> (REGENNET_QKV_X3_NS=1)
run() { python bench.py --respacing ddim5 --no-cpu-baseline --steps 20 --warmup 3 --profile-evals 0 "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])'; }
for r in 1 2 3; do
  echo "round $r: dma $(REGENNET_QKV_X3_DMA=1 run)  rs<2> $(run)  rs<1> $(REGENNET_QKV_X3_NS=1 run)   | B=64: dma $(REGENNET_QKV_X3_DMA=1 run --batch 64)  rs<2> $(run --batch 64)  rs<1> $(REGENNET_QKV_X3_NS=1 run --batch 64)"
done
