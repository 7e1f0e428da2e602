#!/usr/bin/env python3
"""A server-sized batch (N regions of 300 reads x 24 haplotypes in one flat batch) resident on the GPU: ms per step, for kernel timelines of the
multi-region path (rocprofv3 --kernel-trace) in both launch modes.   python tools/multi_region_trace.py [n_regions=4]"""
import json
import sys
import time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from octopus_amd import abi, engine, synth   # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
eng = engine.Engine(abi.Config.default(max_indel_error=16))
rng = np.random.default_rng(1)
batch = synth.batch_from_regions([synth.make_region(rng, 300, 24, B=16, positions="none") for _ in range(n)])
out = np.empty(batch.out_size())
rb = eng.upload(batch)
for _ in range(5):
    rb.run(); rb.wait()
ts = []
for _ in range(20):
    t0 = time.perf_counter(); rb.run(); rb.wait(); ts.append(time.perf_counter() - t0)
tp = []
for _ in range(20):
    t0 = time.perf_counter(); eng.populate(batch, out=out); tp.append(time.perf_counter() - t0)
print(json.dumps({"regions": n, "device_sized": rb.device_sized(), "run_ms_median": sorted(ts)[10] * 1e3, "populate_ms_median": sorted(tp)[10] * 1e3, "stats": rb.stats()}))
