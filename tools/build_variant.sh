# A variant build of the library for same-box A/B runs (tools/ab_many.sh): tools/build_variant.sh NAME SOURCE [-DMACRO=V ...]
# Recompiles ONE source of regennet_amd/csrc with the given defines and links it with the objects of the regular build (build/obj,
# made by `python __graft_entry__.py`) into build/lib_NAME.so. build/ is git-ignored and travels to the GPU box with gpurun.
set -eu
NAME=$1; SRC=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/regennet_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden"
mkdir -p $ROOT/build/var_$NAME
/opt/rocm/bin/hipcc $FLAGS "$@" -c $CS/$SRC -o $ROOT/build/var_$NAME/$SRC.o
OBJS=""
for o in $ROOT/build/obj/*.o; do
  b=$(basename $o)
  if [ "$b" = "$SRC.o" ]; then OBJS="$OBJS $ROOT/build/var_$NAME/$SRC.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc $FLAGS -shared -Wl,--version-script=$CS/exports.map -o $ROOT/build/lib_$NAME.so $OBJS
echo "built build/lib_$NAME.so"
