#!/usr/bin/env python3
"""Golden matrices of INTEGRATION.md's second seam: the reference's own read_assigner.cpp:145-287 (oracle/_ref/libref_assigner.so, built from /root/reference by
`make -C oracle patched`) on the seeded scenarios of tests/check_assigner_patch.py, serial branch (its thread-pool branch shares one model between the pool's threads: see tests/check_assigner_patch.py).
The GPU box compares the patched seam with these (it has no /root/reference and need not run the reference's functions).

    python tests/golden/make_assigner_seam_golden.py
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import check_assigner_patch as ca   # noqa: E402

rng = np.random.default_rng(77)
out = []
for band, ploidy, n_reads, T, span, templates, threads, cap in ca.SCENARIOS:
    sc = ca.scenario(rng, ploidy, n_reads, T, span, templates)
    rc, serial, _ = ca.likelihoods("ref", sc, band, 1, cap)
    assert rc == 0
    out.append(serial.tolist())
ca.GOLDEN.write_text(json.dumps({"source": "reference read_assigner.cpp:145-287 via oracle/ref_assigner_bridge.cpp, scenarios of tests/check_assigner_patch.py (seed 77)",
                                 "matrices": out}))
print(sum(len(r) * len(r[0]) for r in out), "values ->", ca.GOLDEN)
