#!/usr/bin/env python3
"""Run one region-sized resident batch a few times (for rocprofv3 --kernel-trace timelines of the small-batch path)."""
import sys
import time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from octopus_amd import abi, engine, synth   # noqa: E402
eng = engine.Engine(abi.Config.default(max_indel_error=16))
rng = np.random.default_rng(1)
batch = synth.batch_from_regions([synth.make_region(rng, 300, 24, B=16, positions="none")])
rb = eng.upload(batch)
for _ in range(5):
    rb.run(); rb.wait()
time.sleep(0.01)
t0 = time.perf_counter()
rb.run(); rb.wait()
print("run ms", (time.perf_counter() - t0) * 1e3)
