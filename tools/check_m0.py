"""ISA check for direct-to-LDS loads: `global_load_lds_*` takes its LDS base from m0, which is the compiler's register too (it keeps the base of its
own __builtin_amdgcn_global_load_lds there and re-materialises it per basic block) and which LLVM does not honour on an inline-asm clobber list.
For every kernel of a source this reads `hipcc -S` and checks that each global_load_lds is dominated, inside its own basic block, by an
write to m0 (`s_mov_b32 m0, ...` / `s_add_i32 m0, ...`) that no restore has undone - and that an inline-asm sequence which saves m0 restores it before
the block ends.
The compile must also be free of the "reserved registers on the clobber list" warning.
    python tools/check_m0.py rgn_layers.hip [more sources]          exit code 0 = clean"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-S", "--cuda-device-only"]


def check(src):
    path = os.path.join(ROOT, "regennet_amd", "csrc", src)
    p = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + [path, "-o", "-"], capture_output=True, text=True)
    if p.returncode:
        return [f"{src}: compile failed\n{p.stderr[-2000:]}"]
    bad = []
    if "reserved registers on the clobber list" in p.stderr:
        bad.append(f"{src}: the compiler warns about a reserved register on an inline-asm clobber list (m0 must be saved / restored inside the asm)")
    kernel, m0_set, n_loads, saved = None, False, 0, None
    for ln in p.stdout.splitlines():
        t = ln.strip()
        if re.match(r"^[A-Za-z_.$][\w.$]*:", t):                  # a label: new basic block (or a new kernel)
            if saved is not None:
                bad.append(f"{src} {kernel}: m0 saved into {saved} by inline asm and not restored before the block ends")
            if not t.startswith(".L"):
                kernel = t.split(":")[0]
            m0_set, saved = False, None
            continue
        if t.startswith(";") or not t:
            continue
        m = re.match(r"s_mov_b32\s+(\S+),\s*m0\b", t)
        if m:
            saved = m.group(1)
            continue
        m = re.match(r"s_mov_b32\s+m0,\s*(\S+)", t)
        if m:
            m0_set = True
            if saved is not None and m.group(1) == saved:
                saved, m0_set = None, False                        # restored: whatever the compiler had there is back, the asm's base is gone
            continue
        if re.match(r"s_\w+\s+m0\b", t):                          # any other scalar instruction with m0 as its destination (the compiler
            m0_set = True                                          # re-materialises the base of a builtin load as `s_add_i32 m0, sN, imm`)
            continue
        if t.startswith("global_load_lds") or re.match(r"buffer_load_\w+.*\blds\b", t):
            n_loads += 1
            if not m0_set:
                bad.append(f"{src} {kernel}: `{t}` has no write to m0 before it in its basic block")
    print(f"{src}: {n_loads} direct-to-LDS loads checked, {len(bad)} problem(s)")
    return bad


if __name__ == "__main__":
    problems = []
    for s in sys.argv[1:] or ["rgn_layers.hip"]:
        problems += check(s)
    for b in problems:
        print("  " + b)
    sys.exit(1 if problems else 0)
