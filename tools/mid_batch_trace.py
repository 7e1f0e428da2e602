#!/usr/bin/env python3
"""A server-sized batch of the configs[3] stream (its first N regions in one flat batch): where one populate's wall time goes (upload / run / wait / download / free, medians),
and a marker populate at the end for kernel timelines (rocprofv3 --kernel-trace --memory-copy-trace).   python tools/mid_batch_trace.py [N=16]"""
import json
import sys
import time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from octopus_amd import abi, engine, synth   # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
regs = synth.region_stream_shard(seed=42, n_regions=max(n, 64), B=16, positions="none")[:n]
batch = synth.batch_from_regions(regs)
eng = engine.Engine(abi.Config.default(max_indel_error=16))
out = np.empty(batch.out_size())
for _ in range(5):
    eng.populate(batch, out=out)
t = {k: [] for k in ("upload", "run", "wait", "download", "free", "populate")}
for _ in range(15):
    c = time.perf_counter(); rb = eng.upload(batch); t["upload"].append(time.perf_counter() - c)
    c = time.perf_counter(); rb.run(); t["run"].append(time.perf_counter() - c)
    c = time.perf_counter(); rb.wait(); t["wait"].append(time.perf_counter() - c)
    c = time.perf_counter(); rb.download(); t["download"].append(time.perf_counter() - c)
    c = time.perf_counter(); rb.free(); t["free"].append(time.perf_counter() - c)
    c = time.perf_counter(); eng.populate(batch, out=out); t["populate"].append(time.perf_counter() - c)
rb = eng.upload(batch); rb.run(); rb.wait(); st = rb.stats(); ds = rb.device_sized(); rb.free()
time.sleep(0.05)
eng.populate(batch, out=out)            # the last populate of the trace
print(json.dumps({"regions": n, "reads": int(batch.n_reads), "haps": int(batch.n_haps), "device_sized": ds, "ms": {k: round(sorted(v)[len(v) // 2] * 1e3, 4) for k, v in t.items()}, "stats": st}))
