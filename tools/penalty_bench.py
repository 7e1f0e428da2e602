#!/usr/bin/env python3
"""Where the per-haplotype penalty vectors are best made: host threads vs one wave per haplotype (LDS) vs one GPU lane per haplotype.
(a) oct_phmm_penalty_vectors on the host, per haplotype; (b) the 2,000-region stream's ~48 k haplotypes: upload with given vectors vs
vectors generated at upload on the host threads / on the device; (c) a region-sized call (300 reads x 24 haplotypes) from host buffers
with given vs generated vectors."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from octopus_amd import abi, engine, synth   # noqa: E402

res = {}
m = engine.default_error_model()
rng = np.random.default_rng(3)
regions = synth.region_stream_shard(seed=42, n_regions=int(os.environ.get("REGIONS", "2000")), B=16, positions="none")
flat = synth.batch_from_regions(regions)
n_haps = len(flat.hap_offsets) - 1
t0 = time.perf_counter(); vec = engine.penalty_vectors(m, flat.hap_bases, flat.hap_offsets); dt = time.perf_counter() - t0
res["host_entry"] = {"haplotypes": n_haps, "ms": dt * 1e3, "us_per_haplotype_all_threads": dt / n_haps * 1e6, "host_threads": min(16, os.cpu_count() or 1)}
one = flat.hap_bases[:int(flat.hap_offsets[1])]
t0 = time.perf_counter()
for _ in range(200):
    engine.penalty_vectors(m, one, np.asarray([0, len(one)], np.uint32))
res["host_entry"]["us_single_haplotype_call"] = (time.perf_counter() - t0) / 200 * 1e6

eng = engine.Engine(abi.Config.default(max_indel_error=16))
eng.set_error_model(m)
nov = flat.without_penalty_vectors()


def time_upload(b, reps=3):
    rb = eng.upload(b); rb.free()
    t0 = time.perf_counter()
    for _ in range(reps):
        rb = eng.upload(b); rb.free()
    return (time.perf_counter() - t0) / reps * 1e3


res["stream_upload_ms"] = {"given_vectors": time_upload(flat)}
for where in ("host", "device", "lanes"):
    os.environ["OCT_PHMM_PENALTIES"] = where
    res["stream_upload_ms"]["generated_on_" + where] = time_upload(nov)
    rb = eng.upload(nov)
    got = rb.penalty_vectors(); rb.free()
    res["stream_upload_ms"]["equal_to_host_entry_" + where] = all(bool(np.array_equal(a, b)) for a, b in zip(got, vec))
del os.environ["OCT_PHMM_PENALTIES"]

g = synth.make_region(rng, 300, 24, B=16, positions="none")
small = synth.batch_from_regions([g])
small_nov = small.without_penalty_vectors()
out = np.empty(small.out_size())
for name, b in (("given_vectors", small), ("generated_vectors", small_nov)):
    for _ in range(20):
        eng.populate(b, out=out)
    t0 = time.perf_counter()
    for _ in range(200):
        eng.populate(b, out=out)
    res.setdefault("region_call_300x24_ms", {})[name] = (time.perf_counter() - t0) / 200 * 1e3
eng.close()
print(json.dumps(res))
