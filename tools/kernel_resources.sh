# Register / spill / scratch metadata of the kernels of one source: tools/kernel_resources.sh rgn_layers.hip [-DMACRO ...]
# (reads the .amdhsa metadata of `hipcc -S`; the numbers the notes in DESIGN.md quote)
set -eu
SRC=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden "$@" -S --cuda-device-only $ROOT/regennet_amd/csrc/$SRC -o - 2>/dev/null | python3 -c "
import sys, re
txt = sys.stdin.read()
for m in re.finditer(r'- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa.target|\Z)', txt, flags=re.S):
    blk = m.group(0)
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    g = lambda k: (re.search(r'\.' + k + r':\s+(\d+)', blk) or [None, '?'])[1]
    print(f\"{name[:70]:70s} vgpr {g('vgpr_count'):>4} agpr {g('agpr_count'):>3} sgpr {g('sgpr_count'):>3} vgpr_spill {g('vgpr_spill_count'):>3} sgpr_spill {g('sgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size'):>6}\")
"
