#!/usr/bin/env python3
"""Golden results of INTEGRATION.md's third seam: the reference's own read_realigner.cpp:83-155 (oracle/_ref/libref_realigner.so, built from /root/reference by
`make -C oracle patched`) on the seeded scenarios of tests/check_realigner_patch.py: per read its new region, CIGAR and log-likelihood.
The GPU box compares the patched seam with these (it has no /root/reference and need not run the reference's functions).

    python tests/golden/make_realigner_seam_golden.py
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import check_realigner_patch as cr   # noqa: E402

rng = np.random.default_rng(91)
out = []
for band, n_reads, T, Lh, want_ll, use_mapq, cap in cr.SCENARIOS:
    sc = cr.scenario(rng, n_reads, T, Lh, band)
    res = cr.realign("ref", sc, band, want_ll, use_mapq, cap)
    assert res["rc"] == 0
    out.append(res)
res = cr.realign("ref", cr.repeat_scenario(rng), 16)               # reads inside repeats longer than themselves (more tied diagonals than the ABI's 15 slots)
assert res["rc"] == 0
out.append(res)
cr.GOLDEN.write_text(json.dumps({"source": "reference read_realigner.cpp:83-155 via oracle/ref_realigner_bridge.cpp, scenarios of tests/check_realigner_patch.py (seed 91)",
                                 "results": out}))
print(sum(len(r["cigar"]) for r in out), "reads ->", cr.GOLDEN)
