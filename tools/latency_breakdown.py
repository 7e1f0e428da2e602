#!/usr/bin/env python3
"""Where a populate call's wall time goes: upload / run+wait / download / free, for a region-sized batch and the 100k x 128 batch.
OCT_LAT_INT32=1: int32 lanes (use_int_scores); OCT_LAT_SMALL=1: skip the 100k x 128 batch."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from octopus_amd import abi, engine, synth   # noqa: E402

cfg = abi.Config.default(max_indel_error=16, use_int_scores=int(os.environ.get("OCT_LAT_INT32", "0")))
eng = engine.Engine(cfg)
rng = np.random.default_rng(1)
res = {}
for name, batch, reps in (("region_300x24", synth.batch_from_regions([synth.make_region(rng, 300, 24, B=16, positions="none")]), 200),
                          ("1kx64", synth.config_batch("1kx64", seed=42, B=16, positions="none"), 50),
                          ("100kx128", synth.config_batch("100kx128", seed=42, B=16, positions="none"), 3))[:2 if os.environ.get("OCT_LAT_SMALL") else 3]:
    outbuf = np.empty(max(batch.out_size(), 1))
    t = dict(upload=0.0, run=0.0, wait=0.0, download=0.0, free=0.0, populate=0.0)
    for rep in range(reps + 2):
        c = time.perf_counter(); rb = eng.upload(batch); d0 = time.perf_counter() - c
        c = time.perf_counter(); rb.run(); d1 = time.perf_counter() - c
        c = time.perf_counter(); rb.wait(); d2 = time.perf_counter() - c
        c = time.perf_counter(); rb.download(); d3 = time.perf_counter() - c
        c = time.perf_counter(); rb.free(); d4 = time.perf_counter() - c
        c = time.perf_counter(); eng.populate(batch, out=outbuf); d5 = time.perf_counter() - c
        if rep >= 2:
            for k, d in zip(t, (d0, d1, d2, d3, d4, d5)):
                t[k] += d / reps * 1e3
    res[name] = {k: round(v, 4) for k, v in t.items()}
print(json.dumps(res))
