"""Synthetic workloads of BASELINE.json's configs: deterministic, numpy only, no oracle.

One "region" = what one HaplotypeLikelihoodArray::populate call sees: H candidate haplotypes derived from a
base haplotype by a few SNV/indel edits, R Illumina-like reads sampled from them, the per-haplotype penalty
vectors of the default `PCR-free.HiSeq-2500` error model where nothing repeats (gap open 45, extend 3, SNV
prior 125, masks = haplotype rotated by one base) plus planted homopolymers with that model's open penalties
(reference core/models/error/error_model_factory.cpp:231-238), and an inactive-flank state.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from . import abi

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
# PCR-free HiSeq-2500 homopolymer (A/T) gap-open penalties by repeat length (error_model_factory.cpp:234)
HOMOPOLYMER_OPEN = np.array([45, 45, 43, 43, 41, 38, 35, 32, 29, 25, 21, 20, 19, 18, 17, 17, 16, 16, 15, 14], dtype=np.int8)


def _penalties(seq: np.ndarray):
    """Default-model penalty vectors for one haplotype sequence."""
    L = len(seq)
    ge = np.full(L, 3, np.int8)
    # homopolymer runs of three or more bases -> table penalty over the run
    change = np.flatnonzero(np.diff(seq) != 0) + 1
    starts = np.concatenate([[0], change])
    ends = np.concatenate([change, [L]])
    n = ends - starts
    go = np.repeat(np.where(n >= 3, HOMOPOLYMER_OPEN[np.minimum(n, len(HOMOPOLYMER_OPEN) - 1)], np.int8(45)).astype(np.int8), n)
    mask_f = np.roll(seq, 1)    # repeat_based_snv_error_model.cpp:174-178: masks are the haplotype rotated by one base
    mask_r = np.roll(seq, -1)
    pr = np.full(L, 125, np.int8)
    return go, ge, mask_f, pr.copy(), mask_r, pr.copy()


def make_haplotypes(rng: np.random.Generator, H: int, Lh: int, n_homopolymers: int = 3):
    base = BASES[rng.integers(0, 4, Lh)].copy()
    for _ in range(n_homopolymers):
        n = int(rng.integers(8, 16))
        a = int(rng.integers(20, Lh - 20 - n))
        base[a:a + n] = BASES[rng.integers(0, 4)]
    haps = [base]
    maps = [np.arange(Lh + 1)]          # base coordinate -> haplotype coordinate
    for _ in range(H - 1):
        seq = list(base)
        cmap = np.arange(Lh + 1)
        for _e in range(int(rng.integers(1, 4))):
            p = int(rng.integers(60, max(61, Lh - 60)))
            kind = rng.random()
            if kind < 0.7:
                alt = BASES[rng.integers(0, 4)]
                q = int(cmap[p])
                if 0 <= q < len(seq):
                    seq[q] = int(alt)
            elif kind < 0.85:
                n = int(rng.integers(1, 11))
                q = int(cmap[p])
                del seq[q:q + n]
                cmap[p + 1:] = np.maximum(cmap[p + 1:] - n, cmap[p])
            else:
                n = int(rng.integers(1, 11))
                q = int(cmap[p])
                seq[q:q] = [int(b) for b in BASES[rng.integers(0, 4, n)]]
                cmap[p:] += n
        arr = np.array(seq, dtype=np.uint8)
        if len(arr) < Lh:     # padded back to Lh
            arr = np.concatenate([arr, BASES[rng.integers(0, 4, Lh - len(arr))]])
        arr = arr[:Lh]
        haps.append(arr)
        maps.append(np.minimum(cmap, Lh))
    return haps, maps


def make_tree_haplotypes(rng: np.random.Generator, H: int, Lh: int, n_homopolymers: int = 3):
    """Haplotypes the way a caller's haplotype generator makes them: ceil(log2 H) candidate variant sites on one base sequence (80 % SNVs, 10 % deletions,
    10 % insertions of 1-10 bases) and H distinct combinations of their alleles, the all-reference combination first. Two haplotypes differ only at the sites
    a read overlaps, so reads meet haplotypes that equal their own over the read's span - the pairs try_naive_evaluate answers."""
    base = BASES[rng.integers(0, 4, Lh)].copy()
    for _ in range(n_homopolymers):
        n = int(rng.integers(8, 16))
        a = int(rng.integers(20, Lh - 20 - n))
        base[a:a + n] = BASES[rng.integers(0, 4)]
    n_sites = int(np.ceil(np.log2(H))) if H > 1 else 0
    sites = np.sort(rng.choice(np.arange(60, Lh - 60, 12), size=n_sites, replace=False)) if n_sites else np.zeros(0, np.int64)   # >= 12 bases apart
    alleles = []
    for p in sites:
        kind = rng.random()
        if kind < 0.8:
            alleles.append(("snv", int(BASES[(int(np.searchsorted(BASES, base[p])) + int(rng.integers(1, 4))) % 4])))
        elif kind < 0.9:
            alleles.append(("del", int(rng.integers(1, 11))))
        else:
            alleles.append(("ins", BASES[rng.integers(0, 4, int(rng.integers(1, 11)))]))
    patterns = np.arange(H) if H == (1 << n_sites) else np.concatenate([[0], 1 + np.sort(rng.choice((1 << n_sites) - 1, size=H - 1, replace=False))])
    haps, maps = [], []
    for pat in patterns:
        seq, cmap, prev, shift = [], np.arange(Lh + 1), 0, 0
        for k, p in enumerate(sites):
            if not (int(pat) >> k) & 1:
                continue
            p = int(p)
            seq.append(base[prev:p])
            kind, val = alleles[k]
            if kind == "snv":
                seq.append(np.array([val], np.uint8)); prev = p + 1
            elif kind == "del":
                prev = p + val
                cmap[p + 1:] = np.maximum(cmap[p + 1:] - val, cmap[p])
            else:
                seq.append(val); prev = p
                cmap[p:] += len(val)
        seq.append(base[prev:])
        arr = np.concatenate(seq).astype(np.uint8)
        if len(arr) < Lh:     # padded back to Lh
            arr = np.concatenate([arr, BASES[rng.integers(0, 4, Lh - len(arr))]])
        haps.append(arr[:Lh])
        maps.append(np.minimum(cmap, Lh))
    return haps, maps


# Base-quality profiles: (values, probabilities, quality lost over the last 30 cycles).
#   "stress"  SURVEY.md 8d: 2 % of the bases at Q2 and 8 % at Q12 plant ~3 mismatches in every read - nearly every candidate reaches the DP (the GCUPS stress)
#   "hq"      a current Illumina run: 88 % Q37, 8 % Q30, ~92 % of all bases >= Q30 after the tail decay, ~0.35 mismatches per read - most
#             candidates against the read's own haplotype are answered by try_naive_evaluate (pair_hmm.hpp:278-319), the regime real reads live in
Q_PROFILES = {"stress": ([37, 25, 12, 2], [0.7, 0.2, 0.08, 0.02], 20),
              "hq": ([37, 30, 25, 12, 2], [0.88, 0.08, 0.025, 0.012, 0.003], 8)}


def make_reads(rng: np.random.Generator, haps, maps, R: int, T: int, B: int, min_diffs: int = 0, q_values=None, indels_per_read: int = 0,
               q_profile: str = "stress", ploidy: int = 0):
    """R reads of length T sampled from uniformly chosen haplotypes (ploidy > 0: from that many of them, the sample's genotype), Illumina-like qualities and errors."""
    H, Lh = len(haps), len(haps[0])
    src = rng.integers(0, H, R) if not ploidy else rng.choice(H, size=min(ploidy, H), replace=False)[rng.integers(0, min(ploidy, H), R)]
    start = rng.integers(B, Lh - T - B + 1, R)              # start in BASE coordinates = the read's reference begin
    q_vals, q_p, q_decay = Q_PROFILES[q_profile]
    if q_values is None:
        quals = rng.choice(np.array(q_vals, dtype=np.uint8), size=(R, T), p=q_p)
    else:
        quals = rng.integers(q_values[0], q_values[1] + 1, size=(R, T)).astype(np.uint8)
    decay = np.concatenate([np.zeros(T - min(30, T)), np.linspace(0, q_decay, min(30, T))]).astype(np.int64)
    quals = np.clip(quals.astype(np.int64) - decay[None, :], 2, 64).astype(np.uint8)
    first = np.minimum(np.stack(maps)[src, start], Lh - T)
    reads = np.stack(haps)[src[:, None], first[:, None] + np.arange(T)[None, :]]
    err = rng.random((R, T)) < np.power(10.0, -quals.astype(np.float64) / 10.0)
    sub = BASES[rng.integers(0, 4, (R, T))]
    reads = np.where(err, sub, reads)
    indel = np.flatnonzero(rng.random(R) < 0.005) if not indels_per_read else np.repeat(np.arange(R), indels_per_read)
    for r in indel:
        p = int(rng.integers(10, T - 10))
        n = int(rng.integers(1, 4))
        row = list(reads[r])
        if rng.random() < 0.5:
            del row[p:p + n]
            row += [int(b) for b in BASES[rng.integers(0, 4, n)]]
        else:
            row[p:p] = [int(b) for b in BASES[rng.integers(0, 4, n)]]
        reads[r] = np.array(row[:T], dtype=np.uint8)
    if min_diffs:
        for r in range(R):
            idx = rng.choice(T, size=min_diffs, replace=False)
            reads[r, idx] = BASES[(np.searchsorted(BASES, reads[r, idx]) + rng.integers(1, 4, min_diffs)) % 4]
    reverse = (rng.random(R) < 0.5).astype(np.uint8)
    mapq = np.full(R, 60, np.uint8)
    return reads, quals, start.astype(np.int64), reverse, mapq, src


def make_region(rng: np.random.Generator, R: int, H: int, T: int = 150, Lh: int = 300, B: int = 16,
                flank=(40, 40), min_diffs: int = 0, positions: str = "true", q_values=None, indels_per_read: int = 0, q_profile: str = "stress",
                hap_model: str = "edits"):
    """Arrays of one populate() call. positions: 'true' = each read's start mapped through the haplotype's edits
    (a stand-in for the k-mer mapper's output), 'none' = leave mapping to the library. hap_model: 'edits' = every haplotype its own 1-3 random edits
    (SURVEY.md 8d), 'tree' = allele combinations of a few candidate sites with reads from a diploid genotype (make_tree_haplotypes)."""
    haps, maps = make_haplotypes(rng, H, Lh) if hap_model == "edits" else make_tree_haplotypes(rng, H, Lh)
    reads, quals, begin, reverse, mapq, _ = make_reads(rng, haps, maps, R, T, B, min_diffs, q_values, indels_per_read, q_profile,
                                                       ploidy=2 if hap_model == "tree" else 0)
    pos = None
    if positions == "true":
        pos = np.stack([np.minimum(m[begin], Lh - T) for m in maps]).astype(np.uint32)     # [H, R]
    return dict(haps=haps, reads=reads, quals=quals, begin=begin, reverse=reverse, mapq=mapq, flank=flank, pos=pos)


def make_linked_region(rng: np.random.Generator, n_long_reads: int, H: int, Lh: int = 1600, chunk: int = 500, B: int = 16, flank=(100, 100), positions: str = "none") -> dict:
    """An active region the way the reference's own long-read configuration presents it to the likelihood model (resources/configs/PacBioCCS.config:
    max-read-length=500, split-long-reads=true, read-linkage=LINKED, max-indel-errors=16, max-assembly-region-size=1000): every long read that crosses the region is
    cut into 500-base chunks, the chunks inside the region are LINKED reads of one template (one likelihood row = the sum over its chunks,
    haplotype_likelihood_model.cpp:306-320), haplotypes are allele combinations over ~1.6 kb. HiFi-like qualities (Q20-40), substitutions at the quality's rate,
    half of the chunks with a 1-2 base indel error, diploid sample."""
    haps, maps = make_tree_haplotypes(rng, H, Lh)
    hs, ms = np.stack(haps), np.stack(maps)
    genotype = rng.choice(H, size=min(2, H), replace=False)
    reads, quals, begin, rows = [], [], [], [0]
    for _ in range(n_long_reads):
        src = int(genotype[rng.integers(0, len(genotype))])
        s0 = B + int(rng.integers(0, chunk))
        while s0 + chunk + B <= Lh:
            first = int(min(ms[src, s0], Lh - chunk))
            row = hs[src, first:first + chunk + 4].copy()
            if rng.random() < 0.5:
                p = int(rng.integers(10, chunk - 10)); n = int(rng.integers(1, 3))
                row = np.concatenate([row[:p], BASES[rng.integers(0, 4, n)], row[p:]]) if rng.random() < 0.5 else np.concatenate([row[:p], row[p + n:]])
            row = row[:chunk].copy()
            if len(row) < chunk:
                row = np.concatenate([row, BASES[rng.integers(0, 4, chunk - len(row))]])
            q = rng.integers(20, 41, chunk).astype(np.uint8)
            err = rng.random(chunk) < np.power(10.0, -q.astype(np.float64) / 10.0)
            row[err] = BASES[rng.integers(0, 4, int(err.sum()))]
            reads.append(row); quals.append(q); begin.append(s0)
            s0 += chunk
        if len(reads) > rows[-1]:
            rows.append(len(reads))
    R = len(reads)
    g = dict(haps=haps, reads=np.stack(reads), quals=np.stack(quals), begin=np.asarray(begin, np.int64), reverse=(rng.random(R) < 0.5).astype(np.uint8),
             mapq=np.full(R, 60, np.uint8), flank=flank, pos=None, row_off=np.asarray(rows, np.int64))
    if positions == "true":
        g["pos"] = np.stack([np.minimum(m[g["begin"]], Lh - chunk) for m in maps]).astype(np.uint32)
    return g


def linked_stream(seed: int, n_regions: int, B: int = 16) -> List[dict]:
    """`ccs-linked`: a stream of such regions, 30-60 x coverage (long reads per region ~ lognormal around 45), haplotypes ~ min(400, geometric(mean 24)) (max-haplotypes=400)."""
    out = []
    for i in range(n_regions):
        rng = np.random.default_rng([seed, 7, i])
        n = int(np.clip(rng.lognormal(np.log(45), 0.4), 8, 200))
        H = int(min(400, rng.geometric(1 / 24.0)))
        out.append(make_linked_region(rng, n, max(H, 1), Lh=1400 + int(rng.integers(0, 401)), B=B))
    return out


def batch_from_regions(regions: List[dict]) -> abi.Batch:
    """Concatenate regions into one flat C-ABI batch (region tables + flank states + CSR positions)."""
    rb, rq, ro, mq, rv, beg = [], [], [0], [], [], []
    hb, ho, go, ge, mf, pf, mr, pr = [], [0], [], [], [], [], [], []
    reg_rows, reg_haps, flank = [0], [0], []
    pos_off, pos_val = [np.zeros(1, np.uint64)], []
    have_pos = all(g["pos"] is not None for g in regions)
    any_rows = any(g.get("row_off") is not None for g in regions)
    row_off, n_reads_so_far = [0], 0
    total = 0
    for g in regions:
        R, T = g["reads"].shape
        if g.get("read_len") is not None:          # reads of different lengths: row r holds read_len[r] bases, the rest of the row is padding
            keep = np.arange(T)[None, :] < np.asarray(g["read_len"])[:, None]
            rb.append(g["reads"][keep]); rq.append(g["quals"][keep])
            ro.extend((ro[-1] + np.cumsum(g["read_len"])).tolist())
        else:
            rb.append(g["reads"].reshape(-1)); rq.append(g["quals"].reshape(-1))
            ro.extend((ro[-1] + T * (np.arange(R) + 1)).tolist())
        mq.append(g["mapq"]); rv.append(g["reverse"]); beg.append(g["begin"])
        for h in g["haps"]:
            a, b, c, d, e, f = _penalties(h)
            hb.append(h); ho.append(ho[-1] + len(h)); go.append(a); ge.append(b); mf.append(c); pf.append(d); mr.append(e); pr.append(f)
        if any_rows:                               # rows = templates of consecutive reads (g["row_off"], region-relative) or single reads
            ro_g = np.asarray(g["row_off"] if g.get("row_off") is not None else np.arange(R + 1), np.int64)
            row_off.extend((n_reads_so_far + ro_g[1:]).tolist())
            reg_rows.append(reg_rows[-1] + len(ro_g) - 1)
        else:
            reg_rows.append(reg_rows[-1] + R)
        n_reads_so_far += R
        reg_haps.append(reg_haps[-1] + len(g["haps"]))
        flank.append(g["flank"] if g["flank"] is not None else (0, 0))
        if have_pos:
            n = len(g["haps"]) * R
            pos_off.append(total + 1 + np.arange(n, dtype=np.uint64))
            pos_val.append(g["pos"].reshape(-1))
            total += n
    b = abi.Batch(
        read_bases=np.concatenate(rb), read_quals=np.concatenate(rq), read_offsets=np.asarray(ro, np.uint32),
        mapq=np.concatenate(mq), reverse=np.concatenate(rv), read_ref_begin=np.concatenate(beg), row_offsets=np.asarray(row_off, np.uint32) if any_rows else None,
        hap_bases=np.concatenate(hb), hap_offsets=np.asarray(ho, np.uint32), hap_ref_begin=np.zeros(len(hb), np.int64),
        gap_open=np.concatenate(go), gap_extend=np.concatenate(ge), snv_mask_fwd=np.concatenate(mf),
        snv_prior_fwd=np.concatenate(pf), snv_mask_rev=np.concatenate(mr), snv_prior_rev=np.concatenate(pr))
    if len(regions) == 1:
        b.flank = regions[0]["flank"]
    else:
        b.region_row_offsets = np.asarray(reg_rows, np.uint32); b.region_hap_offsets = np.asarray(reg_haps, np.uint32)
        b.region_has_flank = np.asarray([1 if g["flank"] is not None else 0 for g in regions], np.uint8)
        b.region_flank = np.asarray(flank, np.uint32)
    if have_pos:
        b.pos_offsets = np.concatenate(pos_off); b.pos_values = np.concatenate(pos_val).astype(np.uint32)
    return b


def config_region(name: str, seed: int = 42, B: int = 16, positions: str = "true") -> dict:
    """The one region of a BASELINE.json config as a region dict (config_batch flattens it): '1k x 64' (configs[0]/[1]),
    '100k x 128' (configs[2]), 'stress' (every read >= 2 differences), 'long64x8' (configs[4])."""
    rng = np.random.default_rng(seed)
    if name == "1kx64":
        return make_region(rng, 1000, 64, B=B, positions=positions)
    if name == "100kx128":
        return make_region(rng, 100_000, 128, B=B, positions=positions)
    if name == "100kx128-hq":      # the same haplotypes (the generator draws them first), reads of a realistic Illumina quality profile
        return make_region(rng, 100_000, 128, B=B, positions=positions, q_profile="hq")
    if name == "10kx64-hq":
        return make_region(rng, 10_000, 64, B=B, positions=positions, q_profile="hq")
    if name == "100kx128-tree":    # a caller's haplotypes (all combinations of seven candidate alleles), diploid sample, hq reads
        return make_region(rng, 100_000, 128, B=B, positions=positions, q_profile="hq", hap_model="tree")
    if name == "10kx16-tree":
        return make_region(rng, 10_000, 16, B=B, positions=positions, q_profile="hq", hap_model="tree")
    if name == "100kx128-nofast":
        return make_region(rng, 100_000, 128, B=B, min_diffs=2, positions=positions)
    if name == "10kx64":
        return make_region(rng, 10_000, 64, B=B, positions=positions)
    if name == "long64x8":      # BASELINE.json configs[4]: 10 kb reads x 20 kb haplotypes, band 256 (int32 lanes), PacBio-like Q 8-15, indel-rich
        return make_region(rng, 64, 8, T=10_000, Lh=20_000, B=256, flank=(400, 400), positions=positions,
                           q_values=(8, 15), indels_per_read=40)
    if name == "long512x8":     # eight times configs[4]'s reads: enough tasks to put several waves on every SIMD (a throughput figure, not a latency one)
        return make_region(rng, 512, 8, T=10_000, Lh=20_000, B=256, flank=(400, 400), positions=positions, q_values=(8, 15), indels_per_read=40)
    if name == "ccs2048x12":    # eight times the reads of ccs256x12: a batch that fills the chip with one task per 16 lanes (ccs256x12 is 1.3 waves per SIMD)
        g = make_region(rng, 2048, 12, T=14_000, Lh=16_000, B=B, flank=(300, 300), positions=positions, q_values=(20, 40), indels_per_read=6)
        g["read_len"] = rng.integers(10_000, 14_001, 2048).astype(np.int64)
        return g
    if name == "ccs256x12":     # UNSPLIT long reads at the PacBio configuration's band (max-indel-errors=16) with the int32 lanes the realigner's model takes for long reads
        # (option_collation.cpp:1687-1693): 256 HiFi-like reads of 10-14 kb (Q20-40, six indel errors each) against 12 haplotypes of 16 kb. The CALLING path of that
        # configuration never sees this shape - it cuts reads into 500-base linked chunks (linked_stream below) - the realignment path does.
        g = make_region(rng, 256, 12, T=14_000, Lh=16_000, B=B, flank=(300, 300), positions=positions, q_values=(20, 40), indels_per_read=6)
        g["read_len"] = rng.integers(10_000, 14_001, 256).astype(np.int64)
        return g
    if name == "tiny":
        return make_region(rng, 40, 6, B=B, positions=positions)
    raise KeyError(name)


def config_batch(name: str, seed: int = 42, B: int = 16, positions: str = "true") -> abi.Batch:
    return batch_from_regions([config_region(name, seed, B, positions)])


def subset_reads(region: dict, idx) -> dict:
    """The same region (haplotypes, flank state) with only the reads `idx`: every (read, haplotype) result is independent of the
    other reads of the call, so the rows of the sub-region equal the corresponding rows of the full one."""
    idx = np.asarray(idx)
    sub = dict(region)
    for k in ("reads", "quals", "begin", "reverse", "mapq") + (("read_len",) if region.get("read_len") is not None else ()):
        sub[k] = region[k][idx]
    if region["pos"] is not None:
        sub["pos"] = region["pos"][:, idx]
    return sub


def region_stream(seed: int, n_regions: int, B: int = 16, positions: str = "true") -> List[dict]:
    """BASELINE.json configs[3] stand-in: active-region stream, R ~ lognormal(median 300, sigma 0.8) in [20, 5000],
    H ~ min(200, geometric(mean 24)), Lh = 300 + U[0, 200], T = 150."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_regions):
        R = int(np.clip(rng.lognormal(np.log(300), 0.8), 20, 5000))
        H = int(min(200, rng.geometric(1 / 24.0)))
        Lh = 300 + int(rng.integers(0, 201))
        out.append(make_region(rng, R, max(H, 1), Lh=Lh, B=B, positions=positions))
    return out


def stream_region(seed: int, i: int, B: int = 16, positions: str = "true", cap=None, hq: bool = False) -> dict:
    """Region i of the fixed active-region stream `seed`: same shape distribution as region_stream, but every region has its own
    generator state, so any rank can produce exactly its share of ONE stream without generating the rest."""
    rng = np.random.default_rng([seed, i])
    R = int(np.clip(rng.lognormal(np.log(300), 0.8), 20, 5000))
    H = int(min(200, rng.geometric(1 / 24.0)))
    Lh = 300 + int(rng.integers(0, 201))
    if cap is not None:          # (max reads, max haplotypes): toy sizes for the simulator-backed tests
        R, H = min(R, cap[0]), min(H, cap[1])
    if hq:                       # the same shapes with a caller's haplotypes (allele combinations, diploid sample) and a current Illumina quality profile
        return make_region(rng, R, max(H, 1), Lh=Lh, B=B, positions=positions, q_profile="hq", hap_model="tree")
    return make_region(rng, R, max(H, 1), Lh=Lh, B=B, positions=positions)


def _stream_regions(args):
    seed, idx, B, positions, cap = args[:5]
    hq = bool(args[5]) if len(args) > 5 else False
    return [stream_region(seed, i, B=B, positions=positions, cap=cap, hq=hq) for i in idx]


def region_stream_shard(seed: int, n_regions: int, rank: int = 0, world: int = 1, B: int = 16, positions: str = "true", cap=None, workers: int = 1,
                        hq: bool = False) -> List[dict]:
    """BASELINE.json configs[3]: the regions i = rank (mod world) of a stream of n_regions regions (round-robin over the GPUs, no exchange).
    A big shard (the 50,000-region stream of bench.py at N > 1) is generated by `workers` spawned processes: every region has its own generator state,
    so the result does not depend on who makes it."""
    idx = list(range(rank, n_regions, world))
    if workers <= 1 or len(idx) < 4000:
        return _stream_regions((seed, idx, B, positions, cap, hq))
    # plain child interpreters (no multiprocessing: its spawn mode re-imports the parent's __main__, and the parent may hold a HIP context that must not be forked)
    import pickle
    import subprocess
    import sys
    root = str(__import__("pathlib").Path(__file__).resolve().parents[1])
    code = ("import sys, pickle; sys.path.insert(0, %r); from octopus_amd import synth; "
            "a = pickle.load(sys.stdin.buffer); pickle.dump(synth._stream_regions(a), sys.stdout.buffer, protocol=4)" % root)
    chunks = [idx[k::workers] for k in range(workers)]
    procs = [subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE) for _ in chunks]
    import threading
    parts = [None] * len(chunks)
    errors = [None] * len(chunks)

    def talk(k):
        try:
            out, _ = procs[k].communicate(pickle.dumps((seed, chunks[k], B, positions, cap, hq), protocol=4))
            if procs[k].returncode != 0:
                raise RuntimeError(f"region generator child {k} exited with {procs[k].returncode}")
            parts[k] = pickle.loads(out)
        except BaseException as e:      # noqa: BLE001 - re-raised in the caller's thread below: a thread's exception is otherwise lost with the thread
            errors[k] = e
    ths = [threading.Thread(target=talk, args=(k,)) for k in range(len(chunks))]
    [t.start() for t in ths]; [t.join() for t in ths]
    for k, e in enumerate(errors):
        if e is not None or parts[k] is None:
            raise RuntimeError(f"region_stream_shard: generator child {k} of {len(chunks)} failed") from e
    by_index = {}
    for c, regs in zip(chunks, parts):
        by_index.update(zip(c, regs))
    return [by_index[i] for i in idx]


def write_regions_file(path, regions: List[dict]) -> None:
    """The regions as tools/region_calls_bench reads them (`--file`): one self-contained record per region, so that a C++ caller can issue one oct_phmm call per
    region without an interpreter in the way. "OCTR", u32 n; per region u32 {R, H, has_flank, lhs, rhs}, u64 {read bases, haplotype bases}, then read offsets
    u32[R + 1], bases, qualities, mapq u8[R], reverse u8[R], ref_begin i64[R], haplotype offsets u32[H + 1], bases, ref_begin i64[H], and the six penalty vectors."""
    import struct
    with open(path, "wb") as f:
        f.write(b"OCTR" + struct.pack("<I", len(regions)))
        for g in regions:
            R, T = g["reads"].shape
            H = len(g["haps"])
            flank = g["flank"]
            hb = np.concatenate(g["haps"]).astype(np.uint8)
            ho = np.concatenate([[0], np.cumsum([len(h) for h in g["haps"]])]).astype(np.uint32)
            pen = [np.concatenate(x) for x in zip(*[_penalties(h) for h in g["haps"]])]        # go, ge, mask_f, prior_f, mask_r, prior_r
            f.write(struct.pack("<5I2Q", R, H, 1 if flank is not None else 0, flank[0] if flank else 0, flank[1] if flank else 0, R * T, len(hb)))
            for a in ((T * np.arange(R + 1)).astype(np.uint32), g["reads"].astype(np.uint8).reshape(-1), g["quals"].astype(np.uint8).reshape(-1), g["mapq"].astype(np.uint8),
                      g["reverse"].astype(np.uint8), g["begin"].astype(np.int64), ho, hb, np.zeros(H, np.int64), pen[0].astype(np.int8), pen[1].astype(np.int8),
                      pen[2].astype(np.uint8), pen[3].astype(np.int8), pen[4].astype(np.uint8), pen[5].astype(np.int8)):
                f.write(np.ascontiguousarray(a).tobytes())
