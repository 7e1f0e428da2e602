#!/usr/bin/env python3
"""From-host streaming of the 2,000-region batch (BASELINE configs[3] stand-in): one populate from host buffers, split into its phases, and k batches in
flight from k host threads (one handle each).   python tools/stream_e2e.py [in_flight ...]"""
import json
import sys
import threading
import time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from octopus_amd import abi, engine, synth   # noqa: E402
ks = [int(a) for a in sys.argv[1:]] or [1, 2, 3]
cfg = abi.Config.default(max_indel_error=16)
batch = synth.batch_from_regions(synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
eng = engine.Engine(cfg)
rb = eng.upload(batch); rb.run(); rb.wait()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); rb.run(); rb.wait(); ts.append(time.perf_counter() - t0)
resident = rb.download().copy(); rb.free()
res = {"resident_ms": min(ts) * 1e3}
t = {"upload": [], "run": [], "download": []}
for _ in range(3):
    t0 = time.perf_counter(); r2 = eng.upload(batch); t1 = time.perf_counter(); r2.run(); r2.wait(); t2 = time.perf_counter(); r2.download(); t3 = time.perf_counter(); r2.free()
    t["upload"].append(t1 - t0); t["run"].append(t2 - t1); t["download"].append(t3 - t2)
res["split_ms"] = {k: min(v) * 1e3 for k, v in t.items()}
pool = engine.PinnedPool()
locked = pool.batch(batch)
for tag, bt, mk in (("", locked, lambda: pool.empty(batch.out_size(), np.float64)), ("_pageable", batch, lambda: np.empty(batch.out_size()))):
  t = {"upload": [], "run": [], "download": []}
  o1 = mk()
  for _ in range(3):
    t0 = time.perf_counter(); r2 = eng.upload(bt); t1 = time.perf_counter(); r2.run(); r2.wait(); t2 = time.perf_counter(); r2.free()
    t["upload"].append(t1 - t0); t["run"].append(t2 - t1)
  for _ in range(3):
    t0 = time.perf_counter(); eng.populate(bt, out=o1); t["download"].append(time.perf_counter() - t0)
  res["split_ms" + tag] = {"upload": min(t["upload"]) * 1e3, "run": min(t["run"]) * 1e3, "one_populate_call": min(t["download"]) * 1e3}
  for k in ks:
    engs = [eng] + [engine.Engine(cfg) for _ in range(k - 1)]
    outs = [mk() for _ in range(k)]
    for e, o in zip(engs, outs):
        e.populate(bt, out=o)
    n_each = max(2, 36 // k)          # (the pipeline's fill and drain - one upload, one run that overlap with nothing - are under 2 % of 36 batches)
    def work(i):
        for _ in range(n_each):
            engs[i].populate(bt, out=outs[i])
    ths = [threading.Thread(target=work, args=(i,)) for i in range(k)]
    t0 = time.perf_counter(); [x.start() for x in ths]; [x.join() for x in ths]; dt = time.perf_counter() - t0
    res[f"in_flight_{k}{tag}"] = {"ms_per_batch": dt / (k * n_each) * 1e3, "regions_per_s": 2000 * k * n_each / dt, "x_resident": (min(ts) * k * n_each) / dt, "equal": all(np.array_equal(o, resident) for o in outs)}
    for e in engs[1:]:
        e.close()
print(json.dumps(res))
