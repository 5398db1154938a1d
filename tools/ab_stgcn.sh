# Same-box A/B of builds of the library under the recogniser bench: tools/ab_stgcn.sh NAME_A NAME_B ... (build/lib_NAME.so), three alternating rounds, ms per forward (T = 60, T = 150)
export TMPDIR=/tmp
cp regennet_amd/libregennet_hip.so /tmp/lib_keep.so
for r in 1 2 3; do
  for L in "$@"; do
    cp build/lib_$L.so regennet_amd/libregennet_hip.so
    timeout 300 python bench.py --config stgcn --steps 10 --warmup 2 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('  $L', [p['ms_per_forward'] for p in d['per_length']])"
  done
done
cp /tmp/lib_keep.so regennet_amd/libregennet_hip.so
