#!/usr/bin/env python3
"""One-off long randomised parity run on the GPU (tests/check_fuzz.py scenarios): python tools/gpu_fuzz.py [seeds=40] [per_seed=50] [first_seed=1000]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import check_fuzz
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
per = int(sys.argv[2]) if len(sys.argv) > 2 else 50
first = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
t0 = time.time(); n = 0
for s in range(first, first + seeds):
    n += check_fuzz.check_fuzz("gpu", seed=s, n=per, tol=1e-9)
print(f"{n} random scenarios, seeds {first}..{first + seeds - 1} (populate incl. the device mapper's positions; align on the alignable half) equal the oracle on the GPU, {time.time() - t0:.0f} s")
