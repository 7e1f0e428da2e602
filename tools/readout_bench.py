#!/usr/bin/env python3
"""Time the genotype read-out on the resident 100k x 128 matrix: all 8,256 diploid genotypes, and the CPU
oracle on a sample of them, in the same run. Not part of bench.py's headline metric.

    python tools/readout_bench.py [--reads 100000] [--haps 128]
"""
import argparse
import itertools
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from octopus_amd import abi, engine, synth   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=100_000)
ap.add_argument("--haps", type=int, default=128)
ap.add_argument("--cpu-sample", type=int, default=64)
a = ap.parse_args()
rng = np.random.default_rng(42)
batch = synth.batch_from_regions([synth.make_region(rng, a.reads, a.haps, B=16, positions="none")])
eng = engine.Engine(abi.Config.default(max_indel_error=16))
rb = eng.upload(batch)
rb.run(); rb.wait()
res = {"reads": a.reads, "haps": a.haps}
for ploidy, gts in ((2, np.asarray(list(itertools.combinations_with_replacement(range(a.haps), 2)), np.uint32)),
                    (3, np.sort(rng.integers(0, a.haps, (20000, 3)), axis=1).astype(np.uint32)),
                    (4, np.sort(rng.integers(0, a.haps, (20000, 4)), axis=1).astype(np.uint32))):
    sets = [dict(genotypes=gts)]
    rb.genotype_likelihoods(sets)
    t0 = time.perf_counter()
    for _ in range(5):
        got = rb.genotype_likelihoods(sets)
    dt = (time.perf_counter() - t0) / 5
    n_terms = len(gts) * a.reads
    res[f"ploidy{ploidy}"] = {"genotypes": len(gts), "ms": dt * 1e3, "G_read_genotype_terms_per_s": n_terms / dt / 1e9,
                             "matrix_bytes_read_once": a.reads * a.haps * 8}
    try:
        import oracle
        lik = rb.download()
        k = min(a.cpu_sample, len(gts))
        t0 = time.perf_counter()
        want = oracle.genotype_likelihoods(lik, batch.hap_out_offsets(), gts[:k])
        cdt = time.perf_counter() - t0
        res[f"ploidy{ploidy}"]["cpu_1thread_G_terms_per_s"] = k * a.reads / cdt / 1e9
        res[f"ploidy{ploidy}"]["max_rel_diff_vs_oracle"] = float(np.max(np.abs(got[:k] - want) / np.maximum(1, np.abs(want))))
    except ImportError:
        pass
print(json.dumps(res))
