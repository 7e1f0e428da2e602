"""Backend-independent L1 (band kernel) parity checks, run by test_sim_*.py (CPU simulator) and test_gpu_*.py (MI355X)."""
import numpy as np

import oracle
from backends import make_engine
from l1_cases import random_case


def check_golden(backend, golden_records, score_bits=16):
    """The reference's own known-answer vectors through oct_phmm_align_windows, on int16 or int32 lanes
    (the reference checks most vectors on both SSE2PairHMM<B, short> and <B, int>)."""
    by_band = {}
    for rec in golden_records:
        for inst in rec["instantiations"]:
            if inst["score_bits"] == score_bits:
                by_band.setdefault(inst["band"], {})[rec["name"]] = rec
    assert sorted(by_band) == [8, 16, 32]
    n = 0
    for band, recs in by_band.items():
        eng = make_engine(backend, max_indel_error=band, use_int_scores=int(score_bits == 32))
        recs = list(recs.values())
        args = dict(truths=[r["test"]["truth"].encode() for r in recs], targets=[r["test"]["target"].encode() for r in recs],
                    quals=[r["test"]["base_qualities"] for r in recs], gap_open=[r["test"]["gap_open"] for r in recs],
                    gap_extend=None, gap_extend_scalar=recs[0]["test"]["gap_extend"], nuc_prior=recs[0]["test"]["nuc_prior"])
        assert all(r["test"]["gap_extend"] == 1 and r["test"]["nuc_prior"] == 4 for r in recs)
        so = eng.align_windows(traceback=False, **args)
        tb = eng.align_windows(traceback=True, **args)
        for r, a, b in zip(recs, so, tb):
            e = r["expected"]
            assert a["score"] == e["score"], (r["name"], a)
            assert (b["score"], b["first_pos"], b["align1"], b["align2"]) == (e["score"], e["begin"], e["align1"], e["align2"]), (r["name"], b)
            n += 1
        eng.close()
    return n


def check_random(backend, band, n_cases, seed, t_lo=9, t_hi=151, masked=True, with_n=True, q_max=64, junk=False, score_bits=16):
    """Seeded random windows: score-only, traceback strings, first_pos and flank score must equal the oracle's."""
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n_cases):
        c = random_case(rng, band, int(rng.integers(t_lo, t_hi)), with_n=with_n, q_max=q_max)
        if junk:
            T = len(c["target"])
            c["target"] = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), T))
            c["quals"] = np.full(T, q_max, np.uint8)
        L = len(c["truth"])
        c["lhs"], c["rhs"] = int(rng.integers(0, L // 2)), int(rng.integers(0, L // 2))
        cases.append(c)
    eng = make_engine(backend, max_indel_error=band, use_int_scores=int(score_bits == 32))
    kw = dict(truths=[c["truth"] for c in cases], targets=[c["target"] for c in cases], quals=[c["quals"] for c in cases],
              gap_open=[c["gap_open"] for c in cases], gap_extend=[c["gap_extend"] for c in cases], nuc_prior=2)
    if masked:
        kw.update(snv_mask=[c["mask"] for c in cases], snv_prior=[c["prior"] for c in cases])
    so = eng.align_windows(traceback=False, **kw)
    if masked:
        tb = eng.align_windows(traceback=True, lhs_flank=[c["lhs"] for c in cases], rhs_flank=[c["rhs"] for c in cases], **kw)
    else:
        tb = eng.align_windows(traceback=True, **kw)
    eng.close()
    okw = lambda c: dict(snv_mask=c["mask"], snv_prior=c["prior"]) if masked else {}
    for i, c in enumerate(cases):
        a = oracle.align(band, score_bits, c["truth"], c["target"], c["quals"], c["gap_open"], c["gap_extend"], 1, nuc_prior=2, traceback=False, **okw(c))
        b = oracle.align(band, score_bits, c["truth"], c["target"], c["quals"], c["gap_open"], c["gap_extend"], 1, nuc_prior=2, traceback=True, **okw(c))
        assert so[i]["score"] == a["score"], (band, i, so[i], a)
        assert tb[i]["score"] == b["score"] and tb[i]["first_pos"] == b["first_pos"], (band, i, tb[i], b)
        if b["first_pos"] >= 0:
            assert (tb[i]["align1"], tb[i]["align2"]) == (b["align1"], b["align2"]), (band, i)
            if masked:
                f = oracle.flank(band, score_bits, len(c["truth"]), c["lhs"], c["rhs"], c["target"], c["quals"], c["mask"], c["prior"],
                                 c["gap_open"], c["gap_extend"], 2, b["first_pos"], b["align1"], b["align2"])
                assert (tb[i]["flank_score"], tb[i]["mask_size"]) == (f[0], f[1]), (band, i, tb[i], f)
    return len(cases)
