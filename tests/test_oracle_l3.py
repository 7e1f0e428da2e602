"""The oracle's L3 restatement (HaplotypeLikelihoodModel::evaluate / align: the loop over candidate positions with its in-range test,
the original-position rule, the shifted fallback / ShortHaplotypeError, the mapping-quality mixture with cap and trigger, strand-selected
SNV vectors, flank state) against the REFERENCE's own core/models/haplotype_likelihood_model.cpp compiled in place into oracle/_ref
(on stand-in Haplotype / AlignedRead types, see oracle/ref_model_bridge.cpp) - one (read, haplotype) pair per case, exact equality."""
import ctypes as C

import numpy as np
import pytest

import oracle
from octopus_amd import abi

pytestmark = pytest.mark.skipif(not oracle.have_ref(), reason="reference build absent")
BASES = np.frombuffer(b"ACGT", np.uint8)


class Args(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("max_indel_error", "use_int_scores", "use_mapping_quality", "mapping_quality_cap",
                                         "mapping_quality_cap_trigger", "use_flank_state")] + \
               [("hap", C.c_void_p), ("hap_len", C.c_uint32), ("hap_begin", C.c_int64), ("gap_open", C.c_void_p), ("gap_extend", C.c_void_p),
                ("mask_f", C.c_void_p), ("prior_f", C.c_void_p), ("mask_r", C.c_void_p), ("prior_r", C.c_void_p),
                ("has_flank", C.c_int32), ("lhs_flank", C.c_uint32), ("rhs_flank", C.c_uint32),
                ("read", C.c_void_p), ("quals", C.c_void_p), ("read_len", C.c_uint32), ("read_begin", C.c_int64), ("mapq", C.c_uint8), ("reverse", C.c_uint8),
                ("positions", C.c_void_p), ("n_positions", C.c_uint32)]


def random_case(rng, band):
    T = int(rng.integers(10, 100))
    short = rng.random() < 0.12
    Lh = int(rng.integers(T + 2, T + 2 * band)) if short else T + 2 * band + int(rng.integers(0, 160))
    hap = BASES[rng.integers(0, 4, Lh)].copy()
    if rng.random() < 0.3 and Lh > 60:
        hap[20:50] = np.tile(hap[20:26], 5)                           # tandem copies: several plausible positions
    begin = int(rng.integers(0, max(1, Lh - T + 1)))                  # the read's own mapping position: anywhere, also too close to either end
    src = int(rng.integers(0, max(1, Lh - T + 1)))
    read = hap[src:src + T].copy()
    for _ in range(int(rng.integers(0, 5))):
        read[int(rng.integers(0, T))] = BASES[rng.integers(0, 4)]
    quals = rng.integers(2, 61, T).astype(np.uint8)
    n_pos = int(rng.integers(0, 8))
    pos = np.unique(rng.integers(0, Lh + 3, n_pos)).astype(np.uint32) if rng.random() < 0.6 else rng.integers(0, Lh + 3, n_pos).astype(np.uint32)
    if rng.random() < 0.4 and len(pos):
        pos[int(rng.integers(0, len(pos)))] = src                        # the true origin is among the candidates
    if rng.random() < 0.3 and len(pos):
        pos[int(rng.integers(0, len(pos)))] = begin                      # the original position is already mapped
    n = Lh
    return dict(hap=hap, read=read, quals=quals, begin=begin, pos=pos, reverse=bool(rng.integers(0, 2)), mapq=int(rng.integers(0, 256)),
                go=rng.integers(20, 80, n).astype(np.int8), ge=rng.integers(1, 10, n).astype(np.int8),
                mask_f=np.roll(hap, 1), mask_r=np.roll(hap, -1), prior_f=rng.integers(10, 126, n).astype(np.int8), prior_r=rng.integers(10, 126, n).astype(np.int8),
                flank=None if rng.random() < 0.3 else (int(rng.integers(0, Lh // 2 + 1)), int(rng.integers(0, Lh // 2 + 1))))


def random_config(rng, band):
    kw = dict(max_indel_error=band, use_int_scores=int(rng.random() < 0.25), use_mapping_quality=int(rng.random() < 0.8),
              use_flank_state=int(rng.random() < 0.8))
    if rng.random() < 0.4:
        kw.update(mapping_quality_cap=int(rng.integers(10, 200)), mapping_quality_cap_trigger=int(rng.integers(0, 220)))
    return kw


def to_args(c, kw):
    keep = [np.ascontiguousarray(c[k]) for k in ("hap", "go", "ge", "mask_f", "prior_f", "mask_r", "prior_r", "read", "quals", "pos")]
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    cfg = abi.Config.default(**kw)
    a = Args(cfg.max_indel_error, cfg.use_int_scores, cfg.use_mapping_quality, cfg.mapping_quality_cap, cfg.mapping_quality_cap_trigger, cfg.use_flank_state,
             p(keep[0]), len(c["hap"]), 0, p(keep[1]), p(keep[2]), p(keep[3]), p(keep[4]), p(keep[5]), p(keep[6]),
             int(c["flank"] is not None), (c["flank"] or (0, 0))[0], (c["flank"] or (0, 0))[1],
             p(keep[7]), p(keep[8]), len(c["read"]), c["begin"], c["mapq"], int(c["reverse"]), p(keep[9]), len(c["pos"]))
    return a, keep


def batch_of(c):
    hap = dict(seq=bytes(c["hap"]), begin=0, gap_open=c["go"], gap_extend=c["ge"], mask_fwd=c["mask_f"], prior_fwd=c["prior_f"],
               mask_rev=c["mask_r"], prior_rev=c["prior_r"])
    read = dict(seq=bytes(c["read"]), quals=c["quals"], mapq=c["mapq"], reverse=c["reverse"], begin=c["begin"])
    b = abi.Batch.from_lists([read], [hap], flank=c["flank"])
    b.pos_offsets = np.asarray([0, len(c["pos"])], np.uint64)
    b.pos_values = np.ascontiguousarray(c["pos"] if len(c["pos"]) else np.zeros(1, np.uint32), dtype=np.uint32)
    return b


@pytest.mark.parametrize("band", [8, 16])
def test_evaluate_over_positions_equals_the_reference_model(band):
    rng = np.random.default_rng(300 + band)
    R = oracle.ref(); n_short = n_ok = 0
    for _ in range(900):
        c, kw = random_case(rng, band), random_config(rng, band)
        a, keep = to_args(c, kw)
        out, ext = C.c_double(0), C.c_uint32(0)
        rc = R.ref_model_evaluate(C.byref(a), C.byref(out), C.byref(ext))
        got, st, _ = oracle.populate(abi.Config.default(**kw), batch_of(c))
        if rc == 1:
            n_short += 1
            assert st.code == abi.ESHORT_HAPLOTYPE and st.required_extension == ext.value, (st.code, st.required_extension, ext.value)
        else:
            n_ok += 1
            assert st.code == abi.OK and got[0] == out.value, (got[0], out.value, kw, c["begin"], c["pos"])
    assert n_short > 10 and n_ok > 500


@pytest.mark.parametrize("band", [8, 16])
def test_align_over_positions_equals_the_reference_model(band):
    rng = np.random.default_rng(400 + band)
    R = oracle.ref(); seen = 0
    for _ in range(600):
        c, kw = random_case(rng, band), random_config(rng, band)
        a, keep = to_args(c, kw)
        lik, mp, n, ext = C.c_double(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0); ops = (C.c_uint32 * 256)()
        rc = R.ref_model_align(C.byref(a), C.byref(lik), C.byref(mp), ops, 256, C.byref(n), C.byref(ext))
        got, st = oracle.align_batch(abi.Config.default(**kw), batch_of(c), 256)
        if rc == 1:
            assert st.code == abi.ESHORT_HAPLOTYPE and st.required_extension == ext.value
        elif rc == 2:
            assert st.code == abi.EOVERFLOW
        else:
            cig = "".join(f"{ops[i] >> 4}{abi.CIGAR_OPS[ops[i] & 15]}" for i in range(n.value))
            assert st.code == abi.OK
            assert (got["cigar_strings"][0], int(got["mapping_position"][0]), float(got["likelihood"][0])) == (cig, mp.value, lik.value), (kw, c["begin"], c["pos"])
            seen += 1
    assert seen > 300


@pytest.mark.parametrize("indexed", [1, 0])
def test_genotype_model_restatement_equals_the_reference_class(indexed):
    """oracle_genotype_likelihoods vs the reference's own constant_mixture_genotype_likelihood_model.cpp (every ploidy 1-7 and every
    zygosity pattern; both the IndexedHaplotype overloads the callers use and the Haplotype ones)."""
    import itertools
    rng = np.random.default_rng(500 + indexed)
    H, Rr = 6, 57
    lik = -np.abs(rng.normal(3, 6, (H, Rr))); lik[rng.random((H, Rr)) < 0.1] = 0.0; lik[2, 5] = -1.7976931348623157e308
    off = np.arange(H + 1, dtype=np.uint64) * Rr
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    for ploidy in range(1, 8):
        gts = np.asarray(list(itertools.combinations_with_replacement(range(H), ploidy)), np.uint32)
        if len(gts) > 400:
            gts = gts[rng.choice(len(gts), 400, replace=False)]
        want = np.zeros(len(gts)); cols = np.ascontiguousarray(lik)
        oracle.ref().ref_genotype_likelihoods(p(cols), H, Rr, p(gts), len(gts), ploidy, indexed, p(want))
        got = oracle.genotype_likelihoods(lik.reshape(-1), off, gts, (0, Rr))
        if indexed:
            assert np.array_equal(got, want), (ploidy, np.flatnonzero(got != want)[:5])
        else:   # the Haplotype overloads group the same cases slightly differently (e.g. no tetraploid special cases): same value to rounding
            assert np.allclose(got, want, rtol=1e-13, atol=0), ploidy


def check_populate_composed_from_the_reference_pieces(backend, tol=0.0):
    """A whole region through the REFERENCE's own code, piece by piece (its error models -> six vectors, its k-mer mapper -> candidate
    positions, its HaplotypeLikelihoodModel::evaluate per read and haplotype), against the oracle's populate with device-style mapping and
    against the product pipeline on the wave simulator: the matrix the caller would see, from nothing but bases, qualities and positions."""
    from test_oracle_error_models import ref_penalty_vectors
    from backends import make_engine
    from octopus_amd import synth
    rng = np.random.default_rng(9)
    band = 8
    g = synth.make_region(rng, 14, 4, T=60, Lh=200, B=band, flank=(30, 25), positions="none", indels_per_read=0)
    g["mapq"] = rng.integers(5, 70, 14).astype(np.uint8)
    vec = [ref_penalty_vectors(bytes(h)) for h in g["haps"]]                   # reference error models
    haps = [dict(seq=bytes(h), begin=0, gap_open=v[0], gap_extend=v[1], mask_fwd=v[2], prior_fwd=v[3], mask_rev=v[4], prior_rev=v[5])
            for h, v in zip(g["haps"], vec)]
    reads = [dict(seq=bytes(g["reads"][r]), quals=g["quals"][r], mapq=int(g["mapq"][r]), reverse=bool(g["reverse"][r]), begin=int(g["begin"][r]))
             for r in range(14)]
    batch = abi.Batch.from_lists(reads, haps, flank=g["flank"])
    kw = dict(max_indel_error=band)
    R = oracle.ref(); out = (C.c_uint32 * 32)()
    want = np.zeros((4, 14))
    for h in range(4):
        for r in range(14):
            hs, rs = bytes(g["haps"][h]), bytes(g["reads"][r])
            n = R.ref_kmer_map(rs, len(rs), hs, len(hs), 10, out)              # reference k-mer mapper
            c = dict(hap=g["haps"][h], read=g["reads"][r], quals=g["quals"][r], begin=int(g["begin"][r]), pos=np.asarray([out[i] for i in range(n)], np.uint32),
                     reverse=bool(g["reverse"][r]), mapq=int(g["mapq"][r]), go=vec[h][0], ge=vec[h][1], mask_f=vec[h][2], prior_f=vec[h][3],
                     mask_r=vec[h][4], prior_r=vec[h][5], flank=g["flank"])
            a, keep = to_args(c, kw)
            v, ext = C.c_double(0), C.c_uint32(0)
            assert R.ref_model_evaluate(C.byref(a), C.byref(v), C.byref(ext)) == 0   # reference likelihood model
            want[h, r] = v.value
    got, st, _ = oracle.populate(abi.Config.default(**kw), batch)
    assert st.code == abi.OK and np.array_equal(got.reshape(4, 14), want)
    eng = make_engine(backend, **kw)
    dev, dst = eng.populate(batch)
    eng.close()
    assert dst.code == abi.OK and np.max(np.abs(dev.reshape(4, 14) - want)) <= tol


def test_populate_composed_from_the_reference_pieces():
    check_populate_composed_from_the_reference_pieces("sim")
