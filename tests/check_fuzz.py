"""Randomised end-to-end parity: random bands, lane widths, regions, templates, flank states, options and candidate sources; every scenario
must equal the oracle (populate: bit for bit on the sim / within the fp64 bar on the GPU; align: position, CIGAR, likelihood)."""
import numpy as np

import check_align as ca
import check_populate as cp
from octopus_amd import abi, synth


def random_scenario(rng):
    band = int(rng.choice([8, 8, 16, 16, 32, 64]))
    wide = int(rng.random() < 0.3)
    n_regions = int(rng.integers(1, 4))
    regions = []
    for _ in range(n_regions):
        T = int(rng.integers(24, 90)); Lh = max(130, T + 2 * band + int(rng.integers(10, 120)))     # the generator plants its edits in [60, Lh - 60)
        R, H = int(rng.integers(3, 28)), int(rng.integers(1, 6))
        flank = None if rng.random() < 0.25 else (int(rng.integers(0, Lh // 2)), int(rng.integers(0, Lh // 2)))
        g = synth.make_region(rng, R, H, T=T, Lh=Lh, B=band, flank=flank, positions="none", indels_per_read=int(rng.random() < 0.3))
        g["mapq"] = rng.integers(0, 255, R).astype(np.uint8)
        if rng.random() < 0.3:                                   # a few non-ACGT bytes: generic kernels
            g["reads"][rng.integers(0, R), rng.integers(0, T)] = ord("N")
            g["haps"][int(rng.integers(0, H))][int(rng.integers(0, Lh))] = ord("N")
        if rng.random() < 0.3:                                   # reads that hang off the haplotype start: shifted-original fallback
            g["begin"][: max(1, R // 4)] = rng.integers(0, band, max(1, R // 4))
        regions.append(g)
    batch = synth.batch_from_regions(regions)
    templates = rng.random() < 0.4
    if templates:                                                # rows of 1-2 consecutive reads inside a region
        rows, reg_rows, r = [0], [0], 0
        for g in regions:
            end = r + g["reads"].shape[0]
            while r < end:
                r += 2 if (end - r >= 2 and rng.random() < 0.5) else 1
                rows.append(r)
            reg_rows.append(len(rows) - 1)
        batch.row_offsets = np.asarray(rows, np.uint32)
        if len(regions) > 1:
            batch.region_row_offsets = np.asarray(reg_rows, np.uint32)
    host_positions = rng.random() < 0.5
    if host_positions:
        cp.mapper_positions(batch, rng=rng, junk=float(rng.choice([0.0, 0.3, 0.6])))
    cfg = dict(max_indel_error=band, use_int_scores=wide)
    if rng.random() < 0.3: cfg["use_mapping_quality"] = 0
    if rng.random() < 0.3: cfg["use_flank_state"] = 0
    if rng.random() < 0.3: cfg.update(mapping_quality_cap=int(rng.integers(20, 100)), mapping_quality_cap_trigger=int(rng.integers(10, 120)))
    if not host_positions and rng.random() < 0.3: cfg["max_mapping_positions"] = int(rng.integers(1, 15))   # bounds the k-mer mapper's output (kmer_mapper.hpp:120)
    return batch, cfg, templates


def check_fuzz(backend, seed, n, tol=0.0):
    rng = np.random.default_rng(seed)
    done = 0
    for it in range(n):
        batch, cfg, templates = random_scenario(rng)
        cp.compare(backend, batch, tol, **cfg)
        if not templates and it % 2 == 0:
            ca.compare_align(backend, batch, max_cigar_ops=96, **cfg)
        done += 1
    return done
