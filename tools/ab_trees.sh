# Same-box A/B of two whole TREES of this repository (library + host code): tools/ab_trees.sh rounds treeA treeB -- bench flags
# (ab_many.sh swaps builds of the library under ONE host tree; across rounds the C-ABI grew, so an older library needs its own host code.)
# A tree is a directory holding bench.py and a built regennet_amd/libregennet_hip.so, e.g. `git archive <commit> | tar -x -C build/r4_tree`
# + `python __graft_entry__.py` inside it.
set -u
N=$1; shift
TREES=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do TREES+=("$1"); shift; done
[ $# -gt 0 ] && shift
for r in $(seq $N); do
  for T in "${TREES[@]}"; do
    v=$(cd $T && python bench.py --no-cpu-baseline --profile-evals 0 "$@" 2>/dev/null | python -c "import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "$T $v"
  done
done
