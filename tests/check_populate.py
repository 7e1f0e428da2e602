"""Backend-independent parity checks of the whole populate() path against the CPU oracle."""
import numpy as np
import pytest

import oracle
from backends import make_engine
from octopus_amd import abi, synth


def mapper_positions(batch: abi.Batch, max_positions=10, rng=None, junk=0.0):
    """CSR of candidate positions per (haplotype, read) pair from the oracle's k-mer mapper restatement, optionally
    salted with random (possibly out-of-range / duplicate) positions to exercise max_score's bookkeeping."""
    ro, ho = batch.region_tables()
    first = (lambda row: row) if batch.row_offsets is None else (lambda row: int(batch.row_offsets[row]))
    offs, vals = [0], []
    for g in range(len(ro) - 1):
        r0, r1 = first(int(ro[g])), first(int(ro[g + 1]))
        for h in range(int(ho[g]), int(ho[g + 1])):
            hs = bytes(batch.hap_bases[batch.hap_offsets[h]:batch.hap_offsets[h + 1]])
            for r in range(r0, r1):
                rs = bytes(batch.read_bases[batch.read_offsets[r]:batch.read_offsets[r + 1]])
                p = oracle.map_query_to_target(rs, hs, max_positions)
                if rng is not None and rng.random() < junk:
                    extra = [int(x) for x in rng.integers(0, len(hs) + 5, int(rng.integers(1, 4)))]
                    p = sorted(set(p + extra))[:max_positions]
                    if rng.random() < 0.3:
                        p = []
                vals.extend(p)
                offs.append(len(vals))
    batch.pos_offsets = np.asarray(offs, np.uint64)
    batch.pos_values = np.asarray(vals if vals else [0], np.uint32)
    return batch


def assert_device_positions(rb, batch, cfg):
    """map_query_to_target (utils/kmer_mapper.hpp:120-159) for every (haplotype, read) pair: the positions the device mapper
    handed to the rest of the run (oct_phmm_batch_candidate_positions) equal the oracle's list, same order, same truncation."""
    got = rb.candidate_positions()
    S = int(cfg.max_mapping_positions)
    n = 0
    for e, (h, r) in enumerate(batch.read_pairs()):
        hs = bytes(batch.hap_bases[batch.hap_offsets[h]:batch.hap_offsets[h + 1]])
        rs = bytes(batch.read_bases[batch.read_offsets[r]:batch.read_offsets[r + 1]])
        want = oracle.map_query_to_target(rs, hs, S)
        assert got[e] == want, (e, h, r, got[e], want)
        n += 1
    assert n == len(got)
    return n


def compare(backend, batch, tol=0.0, **cfg_kw):
    cfg = abi.Config.default(**cfg_kw)
    want, wst, wstats = oracle.populate(cfg, batch, n_threads=2)
    eng = make_engine(backend, **cfg_kw)
    rb = eng.upload(batch)
    rb.run()
    try:
        got = rb.download()
        code = abi.OK
        st = None
    except Exception as e:  # EngineError
        got, code, st = None, e.code, e.status
    stats = rb.stats()
    if code == abi.OK and batch.pos_offsets is None and batch.n_read_pairs() <= 100000:
        assert_device_positions(rb, batch, cfg)       # the device mapper's own output, pair by pair
    rb.free()
    eng.close()
    assert code == wst.code, (code, wst.code, wst.message)
    if code == abi.ESHORT_HAPLOTYPE:
        assert (st.hap_index, st.read_index, st.required_extension) == (wst.hap_index, wst.read_index, wst.required_extension)
        return None
    assert got.shape == want.shape
    if tol == 0.0:
        bad = np.flatnonzero(got != want)
    else:
        bad = np.flatnonzero(~(np.abs(got - want) <= tol))
    assert bad.size == 0, (bad[:10], got[bad[:10]], want[bad[:10]])
    for k in ("n_pairs", "n_candidates", "n_fast_path", "n_dp_score_only", "n_dp_traceback", "band_cells"):
        assert stats[k] == wstats[k], (k, stats, wstats)
    return stats


def small_region(seed, R=24, H=5, T=60, Lh=150, B=8, flank=(20, 20), ragged=False, with_n=False):
    rng = np.random.default_rng(seed)
    g = synth.make_region(rng, R, H, T=T, Lh=Lh, B=B, flank=flank, positions="none")
    if with_n:
        g["reads"][rng.integers(0, R, 3), rng.integers(0, T, 3)] = ord("N")
        g["haps"][1][int(rng.integers(0, Lh))] = ord("N")
    return g, rng


def check_basic(backend, tol=0.0):
    out = {}
    for B, flank in ((8, (20, 20)), (16, None), (16, (30, 25))):
        g, rng = small_region(11 + B, B=B, T=50, Lh=150, flank=flank)
        batch = mapper_positions(synth.batch_from_regions([g]), rng=rng, junk=0.3)
        out[(B, flank)] = compare(backend, batch, tol, max_indel_error=B)
    s = out[(8, (20, 20))]
    assert s["n_dp_traceback"] > 0 and s["n_fast_path"] > 0
    assert out[(16, None)]["n_dp_traceback"] == 0 and out[(16, None)]["n_dp_score_only"] > 0
    return out


def check_generic_bytes(backend, tol=0.0):
    g, rng = small_region(5, with_n=True)
    batch = mapper_positions(synth.batch_from_regions([g]), rng=rng, junk=0.2)
    return compare(backend, batch, tol, max_indel_error=8)


def check_templates_and_regions(backend, tol=0.0):
    g1, rng = small_region(21, R=12, H=3)
    g2, _ = small_region(22, R=9, H=4, flank=None)
    g3, _ = small_region(23, R=7, H=2, T=40, Lh=120)
    batch = synth.batch_from_regions([g1, g2, g3])
    # templates: rows of 1-2 consecutive reads that never straddle a region
    rows, r = [0], 0
    for R in (12, 9, 7):
        end = r + R
        while r < end:
            r += 2 if (end - r >= 2 and rng.random() < 0.6) else 1
            rows.append(r)
    batch.row_offsets = np.asarray(rows, np.uint32)
    region_rows = [0]
    for lim in (12, 21, 28):
        region_rows.append(rows.index(lim))
    batch.region_row_offsets = np.asarray(region_rows, np.uint32)
    batch = mapper_positions(batch, rng=rng, junk=0.2)
    return compare(backend, batch, tol, max_indel_error=8)


def check_ragged_and_edges(backend, tol=0.0):
    """Variable read lengths (including reads shorter than the band), reads hanging off either haplotype end
    (shifted-original fallback), and a ShortHaplotypeError."""
    rng = np.random.default_rng(77)
    B, Lh = 8, 120
    hap = synth.BASES[rng.integers(0, 4, Lh)]
    haps = [hap.copy(), hap.copy()]
    haps[1][60] = ord("A") if haps[1][60] != ord("A") else ord("C")
    reads, quals, begins = [], [], []
    for T, start in ((5, 30), (7, 50), (33, 20), (64, 10), (80, 20), (40, 0), (40, 3), (40, 78), (40, 75), (100, 5)):
        s = min(start, Lh - T)
        seq = hap[s:s + T].copy()
        if T > 20:
            seq[T // 2] = ord("G") if seq[T // 2] != ord("G") else ord("T")
            seq[T // 3] = ord("G") if seq[T // 3] != ord("G") else ord("T")
        reads.append(seq); quals.append(rng.integers(2, 60, T).astype(np.uint8)); begins.append(start)
    def mk(reads, quals, begins, haps):
        rl = [dict(seq=bytes(r), quals=q, mapq=int(rng.integers(0, 70)), reverse=bool(rng.integers(0, 2)), begin=b)
              for r, q, b in zip(reads, quals, begins)]
        hl = []
        for h in haps:
            go, ge, mf, pf, mr, pr = synth._penalties(h)
            hl.append(dict(seq=bytes(h), begin=0, gap_open=go, gap_extend=ge, mask_fwd=mf, prior_fwd=pf, mask_rev=mr, prior_rev=pr))
        return abi.Batch.from_lists(rl, hl, flank=(10, 10))
    batch = mapper_positions(mk(reads, quals, begins, haps), rng=rng, junk=0.5)
    stats = compare(backend, batch, tol, max_indel_error=8)
    # a read longer than the haplotype can hold with its pads -> ShortHaplotypeError with the reference's extension
    T = Lh - 2 * B + 3
    bad = mapper_positions(mk([hap[:T].copy()], [np.full(T, 30, np.uint8)], [2], haps))
    assert compare(backend, bad, tol, max_indel_error=8) is None
    return stats


def check_mapping_quality_options(backend, tol=0.0):
    g, rng = small_region(31)
    g["mapq"] = rng.integers(0, 255, len(g["mapq"])).astype(np.uint8)
    batch = mapper_positions(synth.batch_from_regions([g]), rng=rng)
    compare(backend, batch, tol, max_indel_error=8, use_mapping_quality=0)
    compare(backend, batch, tol, max_indel_error=8, mapping_quality_cap=60, mapping_quality_cap_trigger=40)
    compare(backend, batch, tol, max_indel_error=8, mapping_quality_cap=30, mapping_quality_cap_trigger=40)   # trigger >= cap is dropped
    compare(backend, batch, tol, max_indel_error=8, use_flank_state=0)


def check_device_kmer_mapper(backend, tol=0.0):
    """positions == NULL: the library maps reads with its own 6-mer voter (utils/kmer_mapper.hpp restated on the device);
    the oracle maps with its CPU restatement. Repeat-rich haplotypes make ties (several equally voted offsets)."""
    out = []
    for seed, B, kw in ((41, 8, {}), (42, 16, dict(R=30, H=4, T=70, Lh=200)), (43, 8, dict(with_n=True))):
        g, rng = small_region(seed, B=B, **kw)
        # plant a tandem repeat and a homopolymer in every haplotype so that many diagonals tie
        for h in g["haps"]:
            a = int(rng.integers(10, len(h) - 50))
            h[a:a + 24] = np.tile(h[a:a + 3], 8)
            h[a + 30:a + 42] = h[a + 30]
        batch = synth.batch_from_regions([g])
        assert batch.pos_offsets is None
        out.append(compare(backend, batch, tol, max_indel_error=B))
    # ragged reads incl. reads shorter than a k-mer (no candidates), several regions
    g1, rng = small_region(44, R=10, H=3)
    g2, _ = small_region(45, R=6, H=2, T=40, Lh=100, flank=None)
    batch = synth.batch_from_regions([g1, g2])
    out.append(compare(backend, batch, tol, max_indel_error=8))
    rng = np.random.default_rng(46)
    hap = synth.BASES[rng.integers(0, 4, 90)]
    go, ge, mf, pf, mr, pr = synth._penalties(hap)
    reads = [dict(seq=bytes(hap[s:s + T]), quals=rng.integers(5, 50, T).astype(np.uint8), mapq=40, reverse=False, begin=s)
             for T, s in ((4, 20), (5, 30), (6, 40), (7, 12), (30, 25))]
    hl = [dict(seq=bytes(hap), begin=0, gap_open=go, gap_extend=ge, mask_fwd=mf, prior_fwd=pf, mask_rev=mr, prior_rev=pr)]
    out.append(compare(backend, abi.Batch.from_lists(reads, hl, flank=None), tol, max_indel_error=8))
    return out


def check_kmer_mapper_positions(backend, seeds=(71, 72, 73, 74, 75, 76)):
    """The device mapper's positions against the oracle's mapper on haplotypes built to tie: pure tandem repeats and homopolymers (dozens of
    diagonals with the same vote count), two copies of one segment (two winners), reads that straddle a copy boundary, random sequence
    (one winner), and max_mapping_positions from 1 to 15 (truncation of the ascending list). Returns the number of pairs checked."""
    n = 0
    for seed in seeds:
        rng = np.random.default_rng(seed)
        T, R = int(rng.integers(30, 100)), 12
        Lh = int(rng.integers(2 * T + 40, 420))
        haps = []
        base = synth.BASES[rng.integers(0, 4, Lh)]
        haps.append(base.copy())
        unit = synth.BASES[rng.integers(0, 4, int(rng.integers(1, 7)))]
        haps.append(np.resize(unit, Lh).copy())                                   # one tandem repeat end to end
        two = base.copy(); seg = int(rng.integers(T // 2, Lh // 2)); two[seg:2 * seg] = two[:seg]; haps.append(two)   # a duplicated segment
        mix = base.copy(); a = int(rng.integers(0, Lh - 60)); mix[a:a + 50] = np.resize(unit, 50); haps.append(mix)
        hom = base.copy(); hom[20:20 + min(80, Lh - 40)] = hom[20]; haps.append(hom)
        reads = []
        for i in range(R):
            src = haps[i % len(haps)]
            s0 = int(rng.integers(0, Lh - T + 1))
            seq = src[s0:s0 + T].copy()
            for _ in range(int(rng.integers(0, 3))): seq[int(rng.integers(0, T))] = synth.BASES[int(rng.integers(0, 4))]
            reads.append(dict(seq=bytes(seq), quals=rng.integers(5, 45, T).astype(np.uint8), mapq=50, reverse=bool(i & 1), begin=int(rng.integers(8, Lh - T - 7))))
        hl = []
        for hb in haps:
            go, ge, mf, pf, mr, pr = synth._penalties(hb)
            hl.append(dict(seq=bytes(hb), begin=0, gap_open=go, gap_extend=ge, mask_fwd=mf, prior_fwd=pf, mask_rev=mr, prior_rev=pr))
        batch = abi.Batch.from_lists(reads, hl, flank=None)
        S = int(rng.choice([1, 2, 3, 10, 15]))
        cfg = abi.Config.default(max_indel_error=8, max_mapping_positions=S)
        eng = make_engine(backend, max_indel_error=8, max_mapping_positions=S)
        rb = eng.upload(batch)
        rb.run()
        try:
            rb.wait()
            n += assert_device_positions(rb, batch, cfg)
        except Exception as e:      # a ShortHaplotypeError of the later stages does not concern the mapper: its positions are still there
            if getattr(e, "code", None) != abi.ESHORT_HAPLOTYPE: raise
        rb.free()
        eng.close()
    return n


def check_int32_lanes(backend, tol=0.0):
    """Config::use_int_scores: the whole populate path on int32 lanes."""
    g, rng = small_region(51, with_n=True)
    batch = mapper_positions(synth.batch_from_regions([g]), rng=rng, junk=0.2)
    a = compare(backend, batch, tol, max_indel_error=8, use_int_scores=1)
    g2, _ = small_region(52, B=16, T=50, Lh=150, flank=(30, 25))
    b = compare(backend, synth.batch_from_regions([g2]), tol, max_indel_error=16, use_int_scores=1)
    return a, b


def check_wide_and_long(backend, tol=0.0, long_T=300, long_Lh=1000, n_reads=6):
    """Bands 128/256 (streaming kernel) and a long-read region whose tables cannot live in LDS (BASELINE.json configs[4] in small):
    int32 lanes, band 256, device k-mer mapping through the big-haplotype mapper when needed."""
    g, rng = small_region(61, R=8, H=2, T=60, Lh=420, B=128, flank=(130, 130))
    a = compare(backend, synth.batch_from_regions([g]), tol, max_indel_error=100)
    g, rng = small_region(62, R=n_reads, H=2, T=long_T, Lh=long_Lh, B=256, flank=(280, 280))
    g["quals"][:] = np.clip(g["quals"], 5, 15)            # PacBio-like low qualities -> many differences
    b = compare(backend, synth.batch_from_regions([g]), tol, max_indel_error=256, use_int_scores=1)
    return a, b


def check_one_shot_populate_streams_slices_back(backend):
    """oct_phmm_populate on a multi-slice batch delivers each slice's rows through the pinned landing zone as soon as its epilogue is
    done; the result must equal the resident path's download (and the oracle)."""
    rng = np.random.default_rng(21)
    batch = synth.batch_from_regions([synth.make_region(rng, 70, 7, B=16, positions="none"),
                                      synth.make_region(rng, 40, 5, B=16, positions="none")])
    cfg = abi.Config.default(max_indel_error=16)
    want, _, _ = oracle.populate(cfg, batch, n_threads=2)
    eng = make_engine(backend, max_indel_error=16)
    got, st = eng.populate(batch)
    assert st.code == abi.OK and np.array_equal(got, want)
    rb = eng.upload(batch); rb.run()
    assert np.array_equal(rb.download(), want)
    rb.free(); eng.close()


def check_long_reads_at_narrow_bands(backend, tol=0.0, T=1400, Lh=3600, n_reads=5):
    """The reference's PacBio configuration keeps band 16 for 10-20 kb reads (resources/configs/PacBioCCS.config: max-indel-errors=16):
    reads and haplotypes far too long for the LDS-resident kernels at bands 8/16/32 go through the streaming kernel with one task per row
    of B lanes, int16 and int32 lanes, score-only and traceback, plus the align path."""
    out = []
    for band, bits, seed in ((16, 1, 71), (16, 0, 72), (8, 1, 73), (32, 0, 74)):
        g, rng = small_region(seed, R=n_reads, H=2, T=T, Lh=Lh, B=band, flank=(Lh // 3, Lh // 3))      # wide inactive flanks: most candidates need the traceback
        g["quals"][:] = np.clip(g["quals"], 5, 20)
        out.append(compare(backend, synth.batch_from_regions([g]), tol, max_indel_error=band, use_int_scores=bits))
    # int32 lanes, the row kernel (k_dp_rows): reads of different lengths in one wave (the rows iterate to the longest one's end, a shorter one stops fetching), reads and
    # haplotypes with an 'N' (generic lists beside the fast-cost ones), bands 16 / 32 / 64; the same bytes as the one-cost streaming kernel (OCT_PHMM_DP_ROWS=0)
    import os
    old = os.environ.get("OCT_PHMM_DP_ROWS")
    try:
        for band, seed, scale in ((16, 81, 1.0), (32, 82, 0.8), (64, 83, 1.6)):
            Tb, Lb = int(T * scale), int(Lh * scale)
            g, rng = small_region(seed, R=n_reads + 3, H=2, T=Tb, Lh=Lb, B=band, flank=(Lb // 3, Lb // 4))
            g["quals"][:] = np.clip(g["quals"], 5, 30)
            g["read_len"] = rng.integers(Tb // 3, Tb + 1, n_reads + 3).astype(np.int64); g["read_len"][0] = Tb
            g["reads"][1, Tb // 5] = ord("N")
            g["haps"][1][Lb // 2] = ord("N")
            batch = synth.batch_from_regions([g])
            os.environ.pop("OCT_PHMM_DP_ROWS", None)
            out.append(compare(backend, batch, tol, max_indel_error=band, use_int_scores=1))
            res = []
            for rows in ("1", "0"):
                os.environ["OCT_PHMM_DP_ROWS"] = rows
                eng = make_engine(backend, max_indel_error=band, use_int_scores=1)
                rb = eng.upload(batch); rb.run(); res.append((rb.download().copy(), rb.stats(), rb.kernel_time_by_kind() if hasattr(rb, "kernel_time_by_kind") else None)); rb.free(); eng.close()
            assert np.array_equal(res[0][0], res[1][0])
            for k in ("n_pairs", "n_candidates", "n_fast_path", "n_dp_score_only", "n_dp_traceback", "band_cells"):
                assert res[0][1][k] == res[1][1][k], k
    finally:
        if old is None:
            os.environ.pop("OCT_PHMM_DP_ROWS", None)
        else:
            os.environ["OCT_PHMM_DP_ROWS"] = old
    return out


def check_empty_batches(backend):
    """No reads, no haplotypes, neither: every entry point returns OK with empty outputs (and a 1 x 1 batch still works after them)."""
    rng = np.random.default_rng(1)
    g = synth.make_region(rng, 6, 2, T=40, Lh=120, B=8, positions="none")
    reads = [dict(seq=bytes(g["reads"][r]), quals=g["quals"][r], mapq=60, reverse=False, begin=int(g["begin"][r])) for r in range(6)]
    haps = []
    for h in g["haps"]:
        go, ge, mf, pf, mr, pr = synth._penalties(h)
        haps.append(dict(seq=bytes(h), begin=0, gap_open=go, gap_extend=ge, mask_fwd=mf, prior_fwd=pf, mask_rev=mr, prior_rev=pr))
    eng = make_engine(backend, max_indel_error=8)
    cfg = abi.Config.default(max_indel_error=8)
    for rr, hh in (([], haps), (reads, []), ([], []), (reads[:1], haps[:1])):
        batch = abi.Batch.from_lists(rr, hh, flank=(5, 5))
        out, st = eng.populate(batch, raise_on_error=False)
        want, wst, _ = oracle.populate(cfg, batch)
        assert st.code == wst.code == abi.OK and np.array_equal(out, want) and len(out) == len(rr) * len(hh)
        aln, ast = eng.align(batch, raise_on_error=False)
        assert ast.code == abi.OK and len(aln["cigar_strings"]) == len(rr) * len(hh)
        rb = eng.upload(batch); rb.run()
        assert len(rb.download()) == len(rr) * len(hh)
        assert len(rb.genotype_likelihoods([])) == 0
        rb.free()
    eng.close()


def check_late_traceback_start(backend, tol=0.0):
    """Windows that overlap only the RIGHT inactive flank run the score-only recurrence until the last tiles and their walk stops when it
    leaves the flank (k_dp late start). Forced on for small batches through the test hook; every case must equal the
    oracle, and the bytes of the same batch with the late start switched off."""
    import os
    rng = np.random.default_rng(1234)
    out = []
    old = os.environ.get("OCT_PHMM_LATE_MIN_PAIRS")
    try:
        for B, T, Lh, flank, with_n, ragged in ((16, 150, 300, (40, 40), False, False), (16, 120, 260, (0, 90), True, False),
                                                (8, 70, 200, (15, 60), False, True), (16, 200, 420, (10, 150), False, True),
                                                (32, 150, 400, (30, 120), False, False), (64, 100, 400, (0, 100), False, False)):
            g = synth.make_region(rng, 20, 4, T=T, Lh=Lh, B=B, flank=flank, positions="none", indels_per_read=1)
            if with_n:
                g["reads"][rng.integers(0, 20, 4), rng.integers(0, T, 4)] = ord("N")
            batch = synth.batch_from_regions([g])
            if ragged:                                              # trim reads to different lengths (Tmin of a wave bounds the switch point)
                rl, hl = [], []
                for r in range(20):
                    n = int(rng.integers(max(B + 4, T // 3), T + 1))
                    rl.append(dict(seq=bytes(g["reads"][r][:n]), quals=g["quals"][r][:n], mapq=int(g["mapq"][r]), reverse=bool(g["reverse"][r]), begin=int(g["begin"][r])))
                for h in g["haps"]:
                    go, ge, mf, pf, mr, pr = synth._penalties(h)
                    hl.append(dict(seq=bytes(h), begin=0, gap_open=go, gap_extend=ge, mask_fwd=mf, prior_fwd=pf, mask_rev=mr, prior_rev=pr))
                batch = abi.Batch.from_lists(rl, hl, flank=flank)
            batch = mapper_positions(batch, rng=rng, junk=0.3)
            os.environ["OCT_PHMM_LATE_MIN_PAIRS"] = "0"
            stats = compare(backend, batch, tol, max_indel_error=B)
            eng = make_engine(backend, max_indel_error=B)
            late, _ = eng.populate(batch)
            os.environ["OCT_PHMM_LATE_MIN_PAIRS"] = "1000000000000"
            plain, _ = eng.populate(batch)
            eng.close()
            assert np.array_equal(late, plain)
            out.append(stats)
    finally:
        if old is None:
            os.environ.pop("OCT_PHMM_LATE_MIN_PAIRS", None)
        else:
            os.environ["OCT_PHMM_LATE_MIN_PAIRS"] = old
    assert sum(s["n_dp_traceback"] for s in out) > 200
    return out


def check_haplotype_beyond_40k_bases(backend, tol=0.0, Lh=45_000, R=6):
    """Haplotypes of 40 k bases and more with the DEVICE k-mer mapper: refused until round 5 (k_kmer_map_big kept a 32-bit counter per diagonal in LDS), served since round 6
    (16-bit counters: a diagonal collects at most one vote per read k-mer and reads are shorter than 32,768 bases) up to the 16-bit bin tables' own limit of 65,535 bases - where
    the library still answers OCT_PHMM_EUNSUPPORTED. Positions pair by pair and the matrix against the oracle."""
    rng = np.random.default_rng(77)
    g = synth.make_region(rng, R, 2, T=150, Lh=Lh, B=16, flank=(40, 40), positions="none")
    batch = synth.batch_from_regions([g])
    stats = compare(backend, batch, tol, max_indel_error=16)
    # 65,536 bases: refused as before
    big = synth.make_region(rng, 2, 1, T=150, Lh=65_600, B=16, flank=(40, 40), positions="none")
    eng = make_engine(backend, max_indel_error=16)
    out, st = eng.populate(synth.batch_from_regions([big]), raise_on_error=False)
    eng.close()
    assert st.code == abi.EUNSUPPORTED, st.code
    return stats


def check_reads_beyond_32k_bases(backend, tol=0.0, T=33_300, Lh=34_100, cases=((16, 1), (256, 1), (64, 0))):
    """Reads of 32,768 bases and more: refused until round 6 (a queued walk event held both window coordinates in 15 bits each), served since (kind | band diagonal | x: 20 bits
    of x - phmm_kernels.hpp, OCT_WALK_EVENT). Wide inactive flanks, so that nearly every candidate needs the traceback and the walks price columns at window coordinates on both
    sides of 32,768; an indel-rich read, so that gap events are queued there too. All three long-read walkers: k_walk_rows (band 16, int32 lanes), k_walk_long (band 256, int32),
    the lockstep walker behind the streaming kernel (band 64, int16 lanes). Device k-mer mapping (haplotypes stay below its 65,536 bases). T + 2B >= 2^20 is still refused."""
    out = []
    for (band, bits), seed in zip(cases, (91, 92, 93)):
        rng = np.random.default_rng(seed)
        g = synth.make_region(rng, 3, 2, T=T, Lh=Lh + 2 * band, B=band, flank=(Lh // 2 - 300, Lh // 2 - 300), positions="none", indels_per_read=6)
        g["quals"][:] = np.clip(g["quals"], 5, 25)
        g["read_len"] = np.asarray([T, T - 977, 32_768 - 2 * band], np.int64)          # (the last one: the old limit's first refused length)
        out.append(compare(backend, synth.batch_from_regions([g]), tol, max_indel_error=band, use_int_scores=bits))
    assert all(st["n_dp_traceback"] > 0 for st in out), out
    big = synth.make_region(np.random.default_rng(94), 1, 1, T=(1 << 20) - 16, Lh=(1 << 20) + 600, B=8, flank=(40, 40), positions="true")
    eng = make_engine(backend, max_indel_error=8)
    _, st = eng.populate(synth.batch_from_regions([big]), raise_on_error=False)
    eng.close()
    assert st.code == abi.EUNSUPPORTED, st.code
    return out


def check_window_pairing(backend, tol=0.0):
    """Big host-sized batches re-order every haplotype's task runs so that the two tasks packed into a lane's halves read ONE haplotype window (k_pair_sort), and
    k_dp runs those segments with pre-packed gap words (dp_groups, PAIRED). Forced on for small batches through the test hook - with the late
    start on and off, one slice and three, ragged reads, several regions, bands 8 / 16 / 32, a traceback budget that cuts the lists into several launches - every case
    must equal the oracle and the bytes of the same batch with pairing off."""
    import os
    rng = np.random.default_rng(4321)
    keys = ("OCT_PHMM_PAIRED", "OCT_PHMM_DEVICE_SIZED", "OCT_PHMM_SLICES", "OCT_PHMM_LATE_MIN_PAIRS", "OCT_PHMM_BP_BUDGET_KB", "OCT_PHMM_PAIRED_MIN_RUN")
    old = {k: os.environ.get(k) for k in keys}
    out = []
    try:
        for B, R, H, T, Lh, flank, slices, late, budget_kb in ((16, 420, 5, 150, 300, (40, 40), "1", "0", None), (16, 300, 7, 150, 330, (40, 40), "3", "0", None),
                                                                (8, 260, 4, 70, 200, (15, 60), "1", "1000000000000", None), (32, 200, 4, 150, 400, (30, 120), "2", "0", None),
                                                                (16, 350, 4, 120, 280, (0, 90), "1", "0", 3000)):
            regs = [synth.make_region(rng, R, H, T=T, Lh=Lh + 13 * i, B=B, flank=flank, positions="none", indels_per_read=0.05) for i in range(2)]
            regs.append(synth.make_region(rng, 9, 3, T=T, Lh=Lh, B=B, flank=flank, positions="none"))      # a region whose classes hold one task each: the odd ones out only
            batch = synth.batch_from_regions(regs)
            os.environ["OCT_PHMM_DEVICE_SIZED"] = "0"; os.environ["OCT_PHMM_SLICES"] = slices; os.environ["OCT_PHMM_LATE_MIN_PAIRS"] = late; os.environ["OCT_PHMM_PAIRED_MIN_RUN"] = "0"
            if budget_kb:
                os.environ["OCT_PHMM_BP_BUDGET_KB"] = str(budget_kb)
            else:
                os.environ.pop("OCT_PHMM_BP_BUDGET_KB", None)
            os.environ["OCT_PHMM_PAIRED"] = "1"
            stats = compare(backend, batch, tol, max_indel_error=B)
            eng = make_engine(backend, max_indel_error=B)
            paired, _ = eng.populate(batch)
            os.environ["OCT_PHMM_PAIRED"] = "0"
            plain, _ = eng.populate(batch)
            eng.close()
            assert np.array_equal(paired, plain)
            out.append(stats)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert sum(s["n_dp_traceback"] for s in out) > 2000 and sum(s["n_dp_score_only"] for s in out) > 1000
    return out


def check_chunked_traceback(backend, tol=0.0):
    """When a slice's traceback tasks do not fit the scratch budget the DP + walk pair runs in chunks of whole workgroups over the same
    scratch. Forced here with a budget of a few task groups (test hook), with and without the late traceback start, populate and align."""
    import os
    keep = {k: os.environ.get(k) for k in ("OCT_PHMM_BP_BUDGET_KB", "OCT_PHMM_LATE_MIN_PAIRS")}
    try:
        for late in ("0", "1000000000000"):
            os.environ["OCT_PHMM_LATE_MIN_PAIRS"] = late
            for B, kb in ((8, 100), (16, 200)):
                os.environ["OCT_PHMM_BP_BUDGET_KB"] = str(kb)
                g, rng = small_region(300 + B, R=40, H=6, T=90, Lh=220, B=B, flank=(25, 60))
                batch = mapper_positions(synth.batch_from_regions([g]), rng=rng, junk=0.3)
                stats = compare(backend, batch, tol, max_indel_error=B)
                assert stats["n_dp_traceback"] > 64
                if late == "0":                                     # align mode keeps every task's backpointers until the winners walk again: one chunk or a clean error
                    eng = make_engine(backend, max_indel_error=B)
                    _, st = eng.align(batch, 64, raise_on_error=False)
                    eng.close()
                    assert st.code == abi.EUNSUPPORTED and b"OCT_PHMM_BP_BUDGET_GB" in bytes(st.message)
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def check_scratch_allocation_failures(backend, tol=0.0):
    """The traceback scratch is the one large allocation of a step. When the device has no room for it (other handles, other processes) a device-sized
    batch falls back to host-sized launches and those halve their chunks until an allocation succeeds; the failed attempts must not leave an error behind
    (the runtime's last-error slot is sticky). Test hook: the first N scratch allocations of a handle fail."""
    import os
    keep = os.environ.get("OCT_PHMM_TEST_FAIL_BP_ALLOCS")
    try:
        for B, fails in ((8, 1), (16, 3)):
            g, rng = small_region(700 + B, R=40, H=6, T=90, Lh=220, B=B, flank=(25, 60))
            batch = synth.batch_from_regions([g])
            os.environ["OCT_PHMM_TEST_FAIL_BP_ALLOCS"] = str(fails)
            stats = compare(backend, batch, tol, max_indel_error=B)
            assert stats["n_dp_traceback"] > 64
            eng = make_engine(backend, max_indel_error=B)
            rb = eng.upload(batch)
            assert rb.device_sized()
            rb.run(); a = rb.download().copy()
            assert not rb.device_sized()                          # the scratch of the device-sized form could not be had
            os.environ.pop("OCT_PHMM_TEST_FAIL_BP_ALLOCS")
            rb2 = eng.upload(batch); rb2.run()
            assert np.array_equal(a, rb2.download())
            rb.free(); rb2.free(); eng.close()
    finally:
        if keep is None:
            os.environ.pop("OCT_PHMM_TEST_FAIL_BP_ALLOCS", None)
        else:
            os.environ["OCT_PHMM_TEST_FAIL_BP_ALLOCS"] = keep


def check_against_reference_array(backend, tol=0.0):
    """The product pipeline against the REFERENCE's own HaplotypeLikelihoodArray::populate (haplotype_likelihood_array.cpp built in place,
    oracle/_ref/libref_array.so): reads and templates, several samples' rows, device k-mer mapping, with no oracle restatement in between."""
    rng = np.random.default_rng(515)
    n_checked = 0
    for B, R, H, T, Lh, flank, tmpl in ((8, 30, 5, 60, 170, (20, 25), False), (16, 40, 6, 100, 260, (40, 40), True), (16, 25, 4, 150, 300, None, False),
                                        (32, 20, 3, 120, 330, (30, 60), True)):
        g = synth.make_region(rng, R, H, T=T, Lh=Lh, B=B, flank=flank, positions="none", indels_per_read=1)
        g["mapq"] = rng.integers(0, 70, R).astype(np.uint8)
        batch = synth.batch_from_regions([g])
        if tmpl:
            rows, r = [0], 0
            while r < R:
                r += min(R - r, int(rng.integers(1, 4)))
                rows.append(r)
            batch.row_offsets = np.asarray(rows, np.uint32)
        cfg = abi.Config.default(max_indel_error=B)
        code, want, _, _, _ = oracle.ref_array_populate(cfg, batch, n_threads=4 if tmpl else 1)
        assert code == 0
        eng = make_engine(backend, max_indel_error=B)
        got, _ = eng.populate(batch)
        eng.close()
        assert got.shape == want.shape
        assert np.max(np.abs(got - want), initial=0.0) <= tol, (B, tmpl)
        n_checked += got.size
    return n_checked


def check_streamed_upload_and_growing_calls(backend, tol=0.0):
    """Inputs larger than the pinned staging buffer stream through its two halves (filled by a few host threads while the DMA drains the
    other half). Forced here with a staging buffer of a few KB (test hook), so item boundaries, padding and the last partial chunk all fall
    inside chunks. Then one handle serves calls of growing and shrinking size (the staging and landing buffers are re-sized independently)."""
    import os
    old = os.environ.get("OCT_PHMM_STAGE_MAX_KB")
    try:
        for kb in ("2", "5", "64"):
            os.environ["OCT_PHMM_STAGE_MAX_KB"] = kb
            check_basic(backend, tol)
            check_templates_and_regions(backend, tol)
            check_device_kmer_mapper(backend, tol)
    finally:
        if old is None:
            os.environ.pop("OCT_PHMM_STAGE_MAX_KB", None)
        else:
            os.environ["OCT_PHMM_STAGE_MAX_KB"] = old
    rng = np.random.default_rng(88)
    cfg = abi.Config.default(max_indel_error=8)
    eng = make_engine(backend, max_indel_error=8)
    for R, H in ((6, 2), (60, 6), (9, 3), (90, 7), (5, 2)):
        batch = synth.batch_from_regions([synth.make_region(rng, R, H, T=50, Lh=150, B=8, flank=(20, 20), positions="none")])
        want, _, _ = oracle.populate(cfg, batch, n_threads=2)
        got, st = eng.populate(batch)
        assert st.code == abi.OK and np.max(np.abs(got - want), initial=0.0) <= tol, (R, H)
    eng.close()



def check_leading_haplotypes_without_reads(backend, tol=0.0):
    """A batch whose first region has haplotypes but no reads, run in several slices: the leading slice holds no pairs, and the read
    hashes every later slice's mapper needs must still be computed (by the first slice that has pairs)."""
    import os
    rng = np.random.default_rng(404)
    g0 = synth.make_region(rng, 1, 3, T=40, Lh=130, B=8, flank=(10, 10), positions="none")
    for k in ("reads", "quals", "begin", "reverse", "mapq"):
        g0[k] = g0[k][:0]
    g1 = synth.make_region(rng, 20, 4, T=50, Lh=150, B=8, flank=(20, 20), positions="none")
    g2 = synth.make_region(rng, 15, 3, T=45, Lh=140, B=8, flank=None, positions="none")
    batch = synth.batch_from_regions([g0, g1, g2])
    old = os.environ.get("OCT_PHMM_SLICES")
    try:
        for n in ("3", "8"):
            os.environ["OCT_PHMM_SLICES"] = n
            stats = compare(backend, batch, tol, max_indel_error=8)
            assert stats["n_pairs"] == 20 * 4 + 15 * 3
    finally:
        if old is None:
            os.environ.pop("OCT_PHMM_SLICES", None)
        else:
            os.environ["OCT_PHMM_SLICES"] = old


def check_shared_pairs(backend, tol=0.0):
    """Exact de-duplication of pairs (k_window_*, k_dedup_match, k_dedup_verify): regions whose haplotypes are copies of one another, copies with a far-away
    edit, copies with an edit in one penalty vector only, and copies of different length. With OCT_PHMM_DEDUP=1 the matrix equals the oracle's
    and the one computed with OCT_PHMM_DEDUP=0 bit for bit, the reference-equivalent counters do not move, and the shared counters show
    that pairs were in fact not recomputed; one and several slices, given and device-mapped positions."""
    import os
    rng = np.random.default_rng(2024)
    regions = []
    shapes = ((40, 6, 50, 170, 8), (70, 9, 60, 220, 8), (33, 3, 45, 150, 8))
    if backend == "sim":                                    # every lane is a coroutine there: the same situations with fewer reads
        shapes = ((22, 6, 50, 170, 8), (34, 9, 60, 220, 8), (16, 3, 45, 150, 8))
    for k, (R, H, T, Lh, B) in enumerate(shapes):
        g = synth.make_region(rng, R, 3, T=T, Lh=Lh, B=B, flank=(20, 20) if k != 2 else None, positions="none")
        haps = list(g["haps"])
        while len(haps) < H:
            src = haps[int(rng.integers(0, 3))].copy()
            kind = len(haps) % 4
            if kind == 1:                                   # an edit near one end: windows far from it stay shared
                src[int(rng.integers(5, 25))] = ord("ACGT"[int(rng.integers(0, 4))])
            elif kind == 2:                                 # shorter haplotype: the same windows, another length
                src = src[:-int(rng.integers(1, 9))]
            elif kind == 3 and len(haps) % 8 == 3:          # an 'N': generic kernels for this one, same windows elsewhere
                src[int(rng.integers(Lh - 30, Lh - 5))] = ord("N")
            haps.append(src)                                # kind 0: an exact copy
        g["haps"] = haps
        regions.append(g)
    # more distinct haplotypes than a read's table holds (kDedupReps = 48), then copies of early and of late ones
    g = synth.make_region(rng, 12 if backend != "sim" else 6, 58, T=45, Lh=160, B=8, flank=(15, 15), positions="none")
    g["haps"] = list(g["haps"]) + [g["haps"][i].copy() for i in (0, 3, 50, 57, 20, 55)]
    regions.append(g)
    batch = synth.batch_from_regions(regions)
    # one haplotype differs from its twin in a penalty vector only
    o = int(batch.hap_offsets[4]); batch.gap_open = batch.gap_open.copy(); batch.gap_open[o + 60:o + 70] -= 3
    import copy
    given = mapper_positions(copy.copy(batch), rng=np.random.default_rng(5), junk=0.2)
    assert batch.pos_offsets is None
    old = {k: os.environ.get(k) for k in ("OCT_PHMM_DEDUP", "OCT_PHMM_SLICES", "OCT_PHMM_DEDUP_HASH_BITS")}
    out = {}
    try:
        for slices in ("1", "3"):
            os.environ["OCT_PHMM_SLICES"] = slices
            for positions in ("device", "given"):
                if backend == "sim" and slices == "3" and positions == "given":
                    continue                                        # (the simulator's budget: three slices with the device's own positions only)
                bt = batch if positions == "device" else given
                res = {}
                for mode in ("0", "1"):
                    os.environ["OCT_PHMM_DEDUP"] = mode
                    eng = make_engine(backend, max_indel_error=8)
                    rb = eng.upload(bt); rb.run(); got = rb.download().copy(); st = rb.stats(); rb.free(); eng.close()
                    res[mode] = (got, st)
                stats = compare(backend, bt, tol, max_indel_error=8)         # against the oracle (OCT_PHMM_DEDUP=1 still set), counters included
                assert np.array_equal(res["0"][0], res["1"][0])
                for k in ("n_pairs", "n_candidates", "n_fast_path", "n_dp_score_only", "n_dp_traceback", "band_cells"):
                    assert res["0"][1][k] == res["1"][1][k] == stats[k], k
                s0, s1 = res["0"][1], res["1"][1]
                assert s0["n_pairs_shared"] == 0 and s0["n_dp_score_only_shared"] + s0["n_dp_traceback_shared"] == 0 and s0["band_cells_shared"] == 0
                assert s1["n_pairs_shared"] > 50 and s1["n_dp_score_only_shared"] + s1["n_dp_traceback_shared"] >= s1["n_pairs_shared"] and s1["band_cells_shared"] > 0
                assert s1["n_dp_score_only_shared"] < s1["n_dp_score_only"] and 0 < s1["n_dp_traceback_shared"] < s1["n_dp_traceback"]
                out[(slices, positions)] = s1
                # hashes only FIND candidates: with both cut to two bits nearly every window and pair collides with an unequal one, every
                # comparison must reject those, and the matrix must not move
                os.environ["OCT_PHMM_DEDUP_HASH_BITS"] = "2"
                eng = make_engine(backend, max_indel_error=8)
                rb = eng.upload(bt); rb.run(); weak = rb.download().copy(); sw = rb.stats(); rb.free(); eng.close()
                del os.environ["OCT_PHMM_DEDUP_HASH_BITS"]
                assert np.array_equal(weak, res["0"][0])
                assert (0 if backend != "sim" else -1) < sw["n_pairs_shared"] <= s1["n_pairs_shared"] and sw["n_dp_score_only"] == s1["n_dp_score_only"]   # (the simulator's smaller regions may lose every share to the collisions)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return out


def check_launch_modes(backend, tol=0.0):
    """A region-sized batch runs with device-sized launches (no mid-step read-back of the task counts, oct_phmm_batch_device_sized); the
    host-sized path that big batches take (forced with OCT_PHMM_DEVICE_SIZED=0, or by a traceback-scratch budget the task bound does not fit)
    must give the same bytes, and both must equal the oracle: packed int16, int32 lanes, the streaming kernels, generic bytes, several regions
    with templates, and the late traceback start."""
    import os
    SW = ("OCT_PHMM_DEVICE_SIZED", "OCT_PHMM_BP_BUDGET_KB", "OCT_PHMM_DSL_TRACE_PER_PAIR", "OCT_PHMM_DSL_MERGE_DP", "OCT_PHMM_HOST_MAPPED",
          "OCT_PHMM_LATE_START")
    keep = {k: os.environ.get(k) for k in SW + ("OCT_PHMM_LATE_MIN_PAIRS",)}
    rng = np.random.default_rng(4711)
    n = repeated = 0
    try:
        cases = []
        shapes = ((16, 150, 300, {}, False), (8, 60, 150, {}, True), (16, 100, 260, dict(use_int_scores=1), False),
                  (128, 120, 600, {}, False), (16, 700, 1500, dict(use_int_scores=1), False), (32, 90, 300, {}, True))
        if backend == "sim":                                       # the simulator runs every lane as a coroutine: the same paths on smaller shapes
            shapes = ((16, 80, 200, {}, False), (8, 50, 130, {}, True), (16, 60, 180, dict(use_int_scores=1), False),
                      (128, 40, 360, {}, False), (16, 260, 700, dict(use_int_scores=1), False), (32, 60, 220, {}, True))
        for B, T, Lh, kw, late in shapes:
            regs = [synth.make_region(rng, int(rng.integers(5, 30 if backend != "sim" else 12)), int(rng.integers(1, 6 if backend != "sim" else 4)), T=T, Lh=Lh, B=B, flank=(20, 30),
                                      positions="none", indels_per_read=1) for _ in range(int(rng.integers(1, 4 if backend != "sim" else 3)))]
            if B == 8:
                for g in regs:
                    g["reads"][rng.integers(0, g["reads"].shape[0], 3), rng.integers(0, T, 3)] = ord("N")
            cases.append((B, kw, late, synth.batch_from_regions(regs)))
        for case_no, (B, kw, late, batch) in enumerate(cases):
            os.environ["OCT_PHMM_LATE_MIN_PAIRS"] = "0" if late else "1000000000000"
            outs = []
            # (round 4's chain of scan launches - "old_chain" - was retired in round 6)
            # against the default (k_scan_fused, a flavour's traceback and late-start lists in one DP launch and one walk); no_late_start: every task writes all its tiles
            modes = ("default", "host", "budget", "overflow", "two_launches", "no_late_start")
            if backend == "sim" and case_no >= 3:                   # (every lane is a coroutine there: the chunked and the multi-region forms on the first three shapes only)
                modes = ("default", "host", "overflow")
            for mode in modes:
                for k in SW:
                    os.environ.pop(k, None)
                if mode == "two_launches":                          # what batches of several regions take: traceback and score-only DP as two launches on two streams instead of one
                    os.environ["OCT_PHMM_DSL_MERGE_DP"] = "0"       # k_dp_pair launch, the tiled scan instead of the one-workgroup one, DMA copies instead of mapped pinned memory
                    os.environ["OCT_PHMM_HOST_MAPPED"] = "0"
                if mode == "host":
                    os.environ["OCT_PHMM_DEVICE_SIZED"] = "0"
                if mode == "no_late_start":
                    os.environ["OCT_PHMM_LATE_START"] = "0"
                if mode == "budget":
                    os.environ["OCT_PHMM_BP_BUDGET_KB"] = "8"
                if mode == "overflow":                              # scratch for one task group: the scan flags the batch, the wait repeats it host-sized
                    os.environ["OCT_PHMM_DSL_TRACE_PER_PAIR"] = "-1"
                eng = make_engine(backend, max_indel_error=B, **kw)
                rb = eng.upload(batch)
                assert rb.device_sized() == (mode in ("default", "overflow", "two_launches", "no_late_start")), (mode, B)
                rb.run(); outs.append(rb.download().copy())
                if mode == "overflow":
                    repeated += 0 if rb.device_sized() else 1
                    assert not rb.device_sized() or rb.stats()["n_dp_traceback"] <= 64        # (one task group per list may hold a tiny case)
                rb.run(); assert np.array_equal(rb.download(), outs[-1])       # a resident batch can be run again
                rb.free()
                one_shot, _ = eng.populate(batch)
                assert np.array_equal(one_shot, outs[-1])
                eng.close()
            assert all(np.array_equal(outs[0], o) for o in outs[1:])
            for k in SW:
                os.environ.pop(k, None)
            compare(backend, batch, tol, max_indel_error=B, **kw)
            n += 1
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert repeated >= 4, repeated
    return n


def check_mapper_mismatch_account(backend, tol=0.0):
    """k_kmer_map_lanes hands k_classify the base mismatches it saw along a pair's one mapped position (DevBatch::pair_mm: none / exactly one at i1 / two or more), so
    that try_naive_evaluate (pair_hmm.hpp:278-319) needs no pass over the bases. Regions of allele-combination haplotypes and high-quality reads (most candidates
    are exact or one mismatch away), plus crafted reads for the branches behind the single mismatch: a substitution inside a homopolymer the read ends in (the
    shifted-suffix tests :305 / :309 with a gap-open penalty below the base quality), a mismatch in the read's first / last six bases (k-mers at the edges), in the
    flank, reads and haplotypes with an N (hash equality is then no base equality: the account must be ignored). Every variant against the oracle (values, counters,
    the mapper's positions) and with the account switched off."""
    import os
    out = []
    for seed, B, T, Lh in ((3, 8, 60, 170), (4, 16, 75, 220), (5, 8, 61, 180), (6, 8, 197, 460), (7, 8, 203, 470)):    # (197: the last three-round read length; 203: four rounds)
        rng = np.random.default_rng(seed)
        g = synth.make_region(rng, 40, 6, T=T, Lh=Lh, B=B, flank=(25, 20), positions="none", q_profile="hq", hap_model="tree")
        haps, reads, quals, begin = g["haps"], g["reads"], g["quals"], g["begin"]
        hp = haps[0]
        run = int(rng.integers(T + B - 6, Lh - T - B + 1))        # every crafted read below stays in range
        for h in haps:
            h[run:run + 10] = ord("A")                              # a homopolymer every haplotype shares (open penalty 21 < Q37)
        quals[:] = np.maximum(quals, 30)
        k = 0

        def put(start, edit, pos_in_read, base=None):
            nonlocal k
            row = hp[start:start + T].copy()
            if edit:
                row[pos_in_read] = base if base is not None else synth.BASES[(int(np.searchsorted(synth.BASES, row[pos_in_read])) + 1) % 4]
            reads[k] = row; begin[k] = start; quals[k] = 37; k += 1
        put(run + 6 - T, True, T - 4, ord("C"))                     # ends inside the homopolymer, substitution inside it: the suffix equals the haplotype shifted by one
        put(run + 8 - T, True, T - 5, ord("G"))
        put(run - 3, True, 4, ord("T"))                             # starts just before it: substitution in the first k-mer only
        for i in (0, 1, 5, 6, T - 1, T - 6, T - 7, T // 2):         # one mismatch at the edges of the k-mer cover
            put(B + 3, True, i)
        put(B, False, 0); put(Lh - T - B, False, 0)                 # exact, at both ends of the range
        put(B + 1, True, 2); reads[k - 1][T - 3] = ord("N")         # a read with an N: its hashes lie
        row = hp[B + 5:B + 5 + T].copy(); row[[7, 30]] = [ord("N"), synth.BASES[(int(np.searchsorted(synth.BASES, row[30])) + 2) % 4]]; reads[k] = row; begin[k] = B + 5; k += 1
        haps[-1][B + 9] = ord("N")                                   # a haplotype with an N (code 0 like A)
        batch = synth.batch_from_regions([g])
        os.environ["OCT_PHMM_LANE_MAPPER"] = "1"                      # the mapper of big batches (one lane per pair): the one that keeps the account
        try:
            on = compare(backend, batch, tol, max_indel_error=B)
            os.environ["OCT_PHMM_MAP_MISMATCHES"] = "0"
            off = compare(backend, batch, tol, max_indel_error=B)
        finally:
            os.environ.pop("OCT_PHMM_MAP_MISMATCHES", None); del os.environ["OCT_PHMM_LANE_MAPPER"]
        wave = compare(backend, batch, tol, max_indel_error=B)         # one wave per pair (what a batch of this size takes by default)
        assert on == off == wave and on["n_fast_path"] > 30, (on, off, wave)
        out.append(on)
    return out


def check_page_locked_caller_buffers(backend, tol=0.0):
    """Big batches whose arrays and `out` live in page-locked memory (oct_phmm_host_alloc) skip the library's staging copies: inputs are copied by the DMA engine
    from the caller's arrays (the library's own small tables still go through the staging halves), results land in `out` itself, slice by slice. Forced on a
    small multi-region batch (OCT_PHMM_STAGE_MAX_KB=2: the streaming upload; OCT_PHMM_PINNED_MIN_KB=0: every array is asked), one slice and three, against the
    oracle and against the same call from pageable arrays; then with only SOME arrays page-locked."""
    import os
    from octopus_amd import engine
    from backends import build_sim
    lib_path = build_sim() if backend == "sim" else None
    regions = [small_region(31 + k, R=10 + 3 * k, H=3 + k, T=40 + 5 * k, Lh=130 + 10 * k)[0] for k in range(3)]
    batch = synth.batch_from_regions(regions)
    cfg = abi.Config.default(max_indel_error=8)
    want, wst, _ = oracle.populate(cfg, batch, n_threads=2)
    assert wst.code == abi.OK
    pool = engine.PinnedPool(lib_path)
    old = {k: os.environ.get(k) for k in ("OCT_PHMM_STAGE_MAX_KB", "OCT_PHMM_PINNED_MIN_KB", "OCT_PHMM_SLICES")}
    n = 0
    try:
        os.environ["OCT_PHMM_STAGE_MAX_KB"] = "2"; os.environ["OCT_PHMM_PINNED_MIN_KB"] = "0"
        locked = pool.batch(batch)
        some = pool.batch(batch); some.read_quals = batch.read_quals.copy(); some.gap_open = batch.gap_open.copy(); some.hap_offsets = batch.hap_offsets.copy()
        for slices in ("1", "3"):
            os.environ["OCT_PHMM_SLICES"] = slices
            eng = make_engine(backend, max_indel_error=8)
            for bt, out in ((locked, pool.empty(batch.out_size(), np.float64)), (some, pool.empty(batch.out_size(), np.float64)), (locked, np.empty(batch.out_size())),
                            (batch, pool.empty(batch.out_size(), np.float64)), (batch, np.empty(batch.out_size()))):
                out[:] = np.nan
                got, st = eng.populate(bt, out=out)
                assert st.code == abi.OK
                bad = np.flatnonzero(~(np.abs(got - want) <= tol)) if tol else np.flatnonzero(got != want)
                assert bad.size == 0, (slices, bad[:5])
                n += 1
            eng.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        pool.close()
    return n


def check_linked_chunks(backend, tol=0.0):
    """Long reads as the reference's PacBioCCS configuration presents them (synth.make_linked_region): linked chunks of one long read form one template (one row, the
    sum over its chunks), several regions with different haplotype lengths in one batch, device k-mer mapping; and ragged reads (synth `read_len`)."""
    out = []
    for seed, (n, H, Lh, chunk, B) in enumerate(((4, 3, 520, 150, 8), (5, 4, 640, 200, 16))):
        rng = np.random.default_rng(500 + seed)
        regs = [synth.make_linked_region(rng, n, H, Lh=Lh + 40 * k, chunk=chunk, B=B, flank=(30, 30)) for k in range(2)]
        batch = synth.batch_from_regions(regs)
        assert batch.row_offsets is not None and batch.n_rows < batch.n_reads
        out.append(compare(backend, batch, tol, max_indel_error=B))
    rng = np.random.default_rng(77)
    g = synth.make_region(rng, 12, 3, T=90, Lh=260, B=8, flank=(20, 20), positions="none")
    g["read_len"] = rng.integers(40, 91, 12)
    out.append(compare(backend, synth.batch_from_regions([g]), tol, max_indel_error=8))
    return out


def check_input_contract(backend):
    """What an upload refuses (OCT_PHMM_EINVAL, nothing launched): a base quality above 127, a negative penalty, an empty read - in a region-sized batch and in one big
    enough (3 MB of qualities, 1.2 M haplotype bases) for the checks to run on several host threads, with the offending byte in the first, a middle and the last chunk.
    The same arrays untouched are accepted."""
    import copy
    rng = np.random.default_rng(77)
    small = synth.batch_from_regions([synth.make_region(rng, 30, 3, B=16, positions="none")])
    g = synth.make_region(rng, 20_000, 1, T=150, Lh=300, B=16, positions="none")
    big = synth.batch_from_regions([g])
    wide = synth.batch_from_regions([synth.make_region(rng, 2, 3_000, T=150, Lh=400, B=16, positions="none")])
    eng = make_engine(backend, max_indel_error=16)
    rb = eng.upload(small); rb.free()
    for batch in (small, big, wide):
        nq, nh = len(batch.read_quals), len(batch.gap_open)
        for where in (0, nq // 2, nq - 1):
            bad = copy.copy(batch); bad.read_quals = batch.read_quals.copy(); bad.read_quals[where] = 128
            with pytest.raises(Exception) as e:
                eng.upload(bad)
            assert e.value.code == abi.EINVAL and "quality" in str(e.value), str(e.value)
        for name in ("gap_open", "gap_extend", "snv_prior_fwd", "snv_prior_rev"):
            for where in (0, nh // 2, nh - 1):
                bad = copy.copy(batch); arr = getattr(batch, name).copy(); arr[where] = -1; setattr(bad, name, arr)
                with pytest.raises(Exception) as e:
                    eng.upload(bad)
                assert e.value.code == abi.EINVAL and "penalty" in str(e.value), (name, where, str(e.value))
        for r in (0, batch.n_reads // 2, batch.n_reads - 1):
            bad = copy.copy(batch); off = batch.read_offsets.copy()
            if r + 1 < batch.n_reads: off[r + 1] = off[r]            # read r empty (read r + 1 takes its bases)
            else: off[r] = off[r + 1]
            bad.read_offsets = off
            with pytest.raises(Exception) as e:
                eng.upload(bad)
            assert e.value.code == abi.EINVAL, str(e.value)
        rb = eng.upload(batch); rb.free()                            # the arrays themselves are fine
    eng.close()


def check_window_tables(backend, tol=0.0):
    """Canonical windows through the per-region table in LDS (k_window_region, the default) and through the tables in global memory (OCT_PHMM_WINDOW_LDS=0): a region whose
    windows fit one pass, one that needs two passes over its keys (more than 8,192 windows) and one with a single haplotype, in one batch; copies, copies with an edit, shorter
    copies. Same matrix as the oracle and as without sharing, and both forms share exactly the same pairs."""
    import os
    rng = np.random.default_rng(4242)
    regions = []
    for R, H, Lh in ((8 if backend == "sim" else 30, 8, 260), (6 if backend == "sim" else 24, 26, 340), (5, 1, 200)):
        g = synth.make_region(rng, R, min(H, 3), T=60, Lh=Lh, B=8, flank=(20, 20), positions="none")
        haps = list(g["haps"])
        while len(haps) < H:
            src = haps[int(rng.integers(0, min(H, 3)))].copy()
            kind = len(haps) % 3
            if kind == 1:
                src[int(rng.integers(5, Lh - 5))] = ord("ACGT"[int(rng.integers(0, 4))])
            elif kind == 2:
                src = src[:-int(rng.integers(1, 9))]
            haps.append(src)
        g["haps"] = haps
        regions.append(g)
    # more haplotypes than the workgroup stages offsets for (kWinHapStage = 1,024): short ones, each a copy of one of three
    g = synth.make_region(rng, 1, 3, T=30, Lh=85, B=8, flank=(10, 10), positions="none")
    g["haps"] = [g["haps"][i % 3].copy() for i in range(1030)]          # 87,550 windows: ELEVEN key classes (a class function with `% 11` lost keys on the GPU: see k_window_region)
    regions.append(g)
    batch = synth.batch_from_regions(regions)
    assert int(batch.hap_offsets[8 + 26] - batch.hap_offsets[8]) > 8192          # the second region: two passes
    assert 10 * 8192 < int(batch.hap_offsets[-1] - batch.hap_offsets[8 + 26 + 1]) <= 11 * 8192
    old = {k: os.environ.get(k) for k in ("OCT_PHMM_DEDUP", "OCT_PHMM_WINDOW_LDS")}
    res = {}
    try:
        for dedup, lds in (("0", "1"), ("1", "1"), ("1", "0")):
            os.environ["OCT_PHMM_DEDUP"] = dedup; os.environ["OCT_PHMM_WINDOW_LDS"] = lds
            eng = make_engine(backend, max_indel_error=8)
            rb = eng.upload(batch); rb.run(); res[(dedup, lds)] = (rb.download().copy(), rb.stats()); rb.free(); eng.close()
        os.environ["OCT_PHMM_DEDUP"] = "1"; os.environ["OCT_PHMM_WINDOW_LDS"] = "1"
        compare(backend, batch, tol, max_indel_error=8)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert np.array_equal(res[("0", "1")][0], res[("1", "1")][0]) and np.array_equal(res[("1", "1")][0], res[("1", "0")][0])
    a, c = res[("1", "1")][1], res[("1", "0")][1]
    assert a["n_pairs_shared"] > 20 and a == c, (a, c)
    return a
