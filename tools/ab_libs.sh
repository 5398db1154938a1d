# Same-box A/B of two builds of the library: tools/ab_libs.sh build/lib_base.so build/lib_new.so [rounds] [bench flags...]
# Swaps regennet_amd/libregennet_hip.so between the two files and runs the default bench line after each swap.
set -u
A=$1; B=$2; N=${3:-3}; shift 3 || shift $#
cp regennet_amd/libregennet_hip.so /tmp/lib_keep.so
for r in $(seq $N); do
  for L in $A $B; do
    cp $L regennet_amd/libregennet_hip.so
    v=$(python bench.py --no-cpu-baseline --steps 2 --warmup 1 --profile-evals 0 --no-row-check "$@" 2>/dev/null | python -c "import sys, json; print(json.loads(sys.stdin.readline())['value'])")
    echo "$L $v"
  done
done
cp /tmp/lib_keep.so regennet_amd/libregennet_hip.so
