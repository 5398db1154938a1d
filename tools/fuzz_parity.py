"""Randomised parity sweep at odd shapes (GPU box): `python tools/fuzz_parity.py [n] [seed]`. The case generator and both checks
live in tests/fuzz_cases.py; `pytest -m gpu` runs a fixed set of 20 of them (tests/test_hip_parity.py), this tool any number."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.fuzz_cases import run_case  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = 0.0
    for case in range(n):
        ok, desc, err, _ = run_case(case, rng)
        worst = max(worst, err if ok else 1.0)
        print(desc + ("" if ok else "   <-- FAIL"), flush=True)
    print(f"worst {worst:.2e}")
    return 0 if worst < 1e-3 else 1


if __name__ == "__main__":
    sys.exit(main())
