#!/usr/bin/env python3
"""How a device batch of N regions of the configs[3] stream scales: resident run (device only) and one oct_phmm_populate from host buffers, per batch and per region.
   python tools/batch_size_scaling.py [N ...]"""
import json
import sys
import time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from octopus_amd import abi, engine, synth   # noqa: E402
ns = [int(a) for a in sys.argv[1:]] or [1, 4, 16, 64, 256, 1024]
regs = synth.region_stream_shard(seed=42, n_regions=max(ns), B=16, positions="none")
eng = engine.Engine(abi.Config.default(max_indel_error=16))
for n in ns:
    batch = synth.batch_from_regions(regs[:n])
    out = np.empty(batch.out_size())
    rb = eng.upload(batch)
    for _ in range(3):
        rb.run(); rb.wait()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); rb.run(); rb.wait(); ts.append(time.perf_counter() - t0)
    st = rb.stats(); rb.free()
    for _ in range(3):
        eng.populate(batch, out=out)
    tp = []
    for _ in range(10):
        t0 = time.perf_counter(); eng.populate(batch, out=out); tp.append(time.perf_counter() - t0)
    run, pop = sorted(ts)[5], sorted(tp)[5]
    print(json.dumps({"regions": n, "pairs": st["n_pairs"], "run_ms": run * 1e3, "populate_ms": pop * 1e3, "run_us_per_region": run / n * 1e6, "populate_us_per_region": pop / n * 1e6,
                      "run_ns_per_pair": run / st["n_pairs"] * 1e9, "populate_ns_per_pair": pop / st["n_pairs"] * 1e9}))
