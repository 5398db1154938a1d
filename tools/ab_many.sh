# Same-box comparison of any number of builds of the library: tools/ab_many.sh rounds lib1.so lib2.so ... [-- bench flags]
set -u
N=$1; shift
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ $# -gt 0 ] && shift
cp regennet_amd/libregennet_hip.so /tmp/lib_keep.so
for r in $(seq $N); do
  for L in "${LIBS[@]}"; do
    cp $L regennet_amd/libregennet_hip.so
    v=$(python bench.py --no-cpu-baseline --steps 2 --warmup 1 --profile-evals 0 --no-row-check "$@" 2>/dev/null | python -c "import sys, json; print(json.loads(sys.stdin.readline())['value'])")
    echo "$L $v"
  done
done
cp /tmp/lib_keep.so regennet_amd/libregennet_hip.so
