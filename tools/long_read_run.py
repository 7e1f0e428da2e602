#!/usr/bin/env python3
"""BASELINE configs[4] (64 x 10 kb reads, 8 x 20 kb haplotypes, band 256, int32 lanes) resident on the GPU, a few timed steps; no CPU check
(tools/long_read_check.py and tests/test_gpu_fullsize.py do that). For rocprofv3 runs of the streaming DP kernels."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from octopus_amd import abi, engine, synth   # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
batch = synth.config_batch("long64x8", seed=42, B=256, positions="none")
eng = engine.Engine(abi.Config.default(max_indel_error=256, use_int_scores=1))
eng.set_timing(True)
rb = eng.upload(batch)
rb.run(); rb.wait()
ts = []
for _ in range(steps):
    t0 = time.perf_counter(); rb.run(); rb.wait(); ts.append(time.perf_counter() - t0)
st = rb.stats()
print(json.dumps({"ms": min(ts) * 1e3, "gcups": st["band_cells"] / min(ts) / 1e9, "dp_kernel_ms_by_kind": rb.kernel_time_by_kind(), "device_sized": rb.device_sized(), "stats": st}))
rb.free(); eng.close()
