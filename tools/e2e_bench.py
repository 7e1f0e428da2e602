#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) timings of the C ABI as a drop-in caller sees it: oct_phmm_populate = upload + run + download from
host buffers, (a) the 100k x 128 batch, (b) one call per active region from T host threads with one handle each (the reference's
calling pattern), (c) the same regions as one flat multi-region batch. Not part of bench.py's headline metric.

    python tools/e2e_bench.py [--regions 400] [--threads 1 4 8 16]
"""
import argparse
import json
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from octopus_amd import abi, engine, synth   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--regions", type=int, default=400)
ap.add_argument("--threads", type=int, nargs="+", default=[1, 4, 8, 16])
ap.add_argument("--skip-big", action="store_true")
a = ap.parse_args()
cfg = abi.Config.default(max_indel_error=16)
res = {}
if not a.skip_big:
    eng = engine.Engine(cfg)
    big = synth.config_batch("100kx128", seed=42, B=16, positions="none")
    outbuf = np.empty(big.out_size())
    eng.populate(big, out=outbuf)
    t0 = time.perf_counter()
    for _ in range(3):
        eng.populate(big, out=outbuf)
    dt = (time.perf_counter() - t0) / 3
    rb = eng.upload(big); rb.run(); rb.wait()
    t0 = time.perf_counter()
    for _ in range(3):
        rb.run(); rb.wait()
    dres = (time.perf_counter() - t0) / 3
    st = rb.stats(); rb.free(); eng.close()
    res["100kx128"] = {"populate_from_host_ms": dt * 1e3, "resident_run_ms": dres * 1e3,
                       "host_bytes_in": int(big.read_bases.nbytes * 2 + big.hap_bases.nbytes * 7), "host_bytes_out": big.out_size() * 8,
                       "GCUPS_pcie_inclusive": st["band_cells"] / dt / 1e9, "GCUPS_resident": st["band_cells"] / dres / 1e9}
regions = synth.region_stream(seed=42, n_regions=a.regions, B=16, positions="none")
batches = [synth.batch_from_regions([g]) for g in regions]
pairs = sum(b.n_read_pairs() for b in batches)
for T in a.threads:
    engs = [engine.Engine(cfg) for _ in range(T)]
    def work(t, warm):
        for i in range(t, len(batches) if not warm else min(len(batches), 4 * T), T):
            engs[t].populate(batches[i])
    for warm in (True, False):
        ths = [threading.Thread(target=work, args=(t, warm)) for t in range(T)]
        t0 = time.perf_counter()
        [x.start() for x in ths]; [x.join() for x in ths]
        dt = time.perf_counter() - t0
    [e.close() for e in engs]
    res[f"per_region_calls_{T}_threads"] = {"regions_per_s": len(batches) / dt, "M_loglik_per_s": pairs / dt / 1e6, "ms_per_call": dt / len(batches) * T * 1e3}
eng = engine.Engine(cfg)
flat = synth.batch_from_regions(regions)
outbuf = np.empty(flat.out_size())
eng.populate(flat, out=outbuf)
t0 = time.perf_counter()
for _ in range(3):
    eng.populate(flat, out=outbuf)
dt = (time.perf_counter() - t0) / 3
rb = eng.upload(flat); rb.run(); rb.wait()
t0 = time.perf_counter()
for _ in range(3):
    rb.run(); rb.wait()
dres = (time.perf_counter() - t0) / 3
t0 = time.perf_counter()
rb2 = eng.upload(flat)
dup = time.perf_counter() - t0
rb.free(); rb2.free()
eng.close()
res["flat_multi_region_batch"] = {"regions": len(regions), "regions_per_s": len(regions) / dt, "M_loglik_per_s": pairs / dt / 1e6, "populate_from_host_ms": dt * 1e3,
                                  "resident_run_ms": dres * 1e3, "upload_ms": dup * 1e3,
                                  "host_bytes_in": int(flat.read_bases.nbytes * 2 + flat.hap_bases.nbytes * 7), "host_bytes_out": flat.out_size() * 8}
print(json.dumps(res))
