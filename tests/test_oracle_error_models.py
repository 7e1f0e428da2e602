"""The penalty-vector restatement (test infrastructure): repeat extraction pinned list-for-list against the reference's own
tandem library built in place, and the two error models' behaviour on constructed haplotypes (default PCR-free HiSeq-2500 tables,
error_model_factory.cpp:231-238 / :488-495)."""
import numpy as np
import pytest

import oracle

AT = [45, 45, 43, 43, 41, 38, 35, 32, 29, 25, 21, 20, 19, 18, 17, 17, 16, 16, 15, 14, 14, 13, 12, 12, 11, 10, 9, 9, 8, 7, 7, 7, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 5]
CG = [45, 45, 45, 41, 39, 34, 30, 24, 21, 18, 15, 13, 12, 10, 8, 7, 7, 6, 6, 6, 6, 6, 6, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 3]
DI = [45, 45, 42, 40, 35, 29, 26, 24, 22, 21, 20, 19, 18, 18, 17, 17, 16, 16, 15, 15, 15, 14, 13, 13, 12, 12, 11, 11, 10, 10, 9, 9, 9, 7, 7, 7, 6, 6, 5, 4, 4, 4, 4, 4, 4, 4, 4, 4, 3]
TRI = [45, 45, 40, 36, 30, 28, 26, 25, 23, 22, 22, 22, 21, 21, 20, 20, 20, 18, 17, 16, 14, 14, 14, 14, 12, 11, 11, 11, 10, 10, 10, 7, 7, 7, 4, 4, 4, 4, 4, 4, 4, 3]
SNV = [[125, 125, 60, 55, 50, 30, 20, 15, 12, 12, 10, 10, 10, 10, 8, 7, 6, 6, 6, 6, 6, 6, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 4, 4, 3, 3, 3, 3, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1],
       [125, 125, 60, 60, 52, 52, 38, 38, 22, 22, 17, 17, 15, 15, 13, 13, 10, 10, 10, 10, 8, 8, 7, 6, 6, 6, 6, 6, 6, 5, 5, 5, 5, 4, 4, 4, 3, 3, 3, 3, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1],
       [125, 125, 125, 55, 55, 55, 40, 40, 40, 25, 25, 25, 19, 19, 19, 11, 11, 11, 9, 9, 9, 7, 7, 6, 6, 6, 6, 6, 6, 5, 5, 5, 5, 4, 4, 4, 3, 3, 3, 3, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1]]


def model():
    return oracle.ErrorModel.make(AT, CG, DI, TRI, SNV)


def random_sequence(rng, n, alphabet=b"ACGT"):
    s = bytearray(alphabet[i] for i in rng.integers(0, len(alphabet), n))
    for _ in range(int(rng.integers(0, 8))):
        p, reps = int(rng.integers(1, 7)), int(rng.integers(2, 12))
        motif = bytes(alphabet[i] for i in rng.integers(0, len(alphabet), p))
        a = int(rng.integers(0, max(1, n - p * reps)))
        s[a:a + p * reps] = motif * reps
    return bytes(s[:n])


@pytest.mark.skipif(not oracle.have_ref(), reason="reference build absent")
@pytest.mark.parametrize("periods", [(1, 5), (1, 3), (1, 1), (2, 3), (1, 4), (2, 6)])
def test_tandem_restatement_equals_the_reference_library(periods):
    """Same repeats in the same order as lib/tandem - including its quirks (runs that touch the end of the string, period == max_period)."""
    rng = np.random.default_rng(periods[0] * 10 + periods[1])
    for it in range(400):
        s = random_sequence(rng, int(rng.integers(1, 400)), b"ACGT" if it % 5 else b"AC")
        assert oracle.tandem_repeats(s, *periods) == oracle.tandem_repeats(s, *periods, backend="ref"), s


def test_tandem_known_answers():
    assert oracle.tandem_repeats(b"ACGTAAAAAAGCT", 1, 5) == [(4, 6, 1)]
    assert oracle.tandem_repeats(b"ACACACACAC", 1, 3) == [(0, 10, 2)]
    assert oracle.tandem_repeats(b"ACACACACAC", 1, 5) == []                       # the Lempel-Ziv path drops a run that reaches the end
    assert oracle.tandem_repeats(b"AACAACAACAATT", 1, 3) == [(0, 2, 1), (1, 10, 3), (3, 2, 1), (6, 2, 1), (9, 2, 1), (11, 2, 1)]
    assert oracle.tandem_repeats(b"", 1, 5) == [] and oracle.tandem_repeats(b"A", 1, 5) == []


def test_indel_model_defaults_and_homopolymer_tables():
    m = model()
    go, ge, mf, pf, mr, pr = oracle.penalty_vectors(m, b"GCAGTATCATGATCGTGAGCACTCGTACGC")     # no exact repeat of any period <= 5
    assert set(go.tolist()) == {45} and set(ge.tolist()) == {3} and set(pf.tolist()) == {125} and set(pr.tolist()) == {125}
    seq = b"GCAGTATC" + b"A" * 12 + b"TGATCGTGAG" + b"C" * 9 + b"ACTCGTACGC"
    go, ge, *_ = oracle.penalty_vectors(m, seq)
    assert set(go[8:20].tolist()) == {AT[12]} and set(ge[8:20].tolist()) == {7}   # A x 12: open table[12] = 19, extension table[12] = 7
    assert set(go[30:39].tolist()) == {CG[9]} and set(ge[30:39].tolist()) == {6}  # C x 9
    assert go[7] == 45 and go[20] == 45
    # CG dinucleotide repeats get the documented -2 (basic_repeat_based_indel_error_model.cpp:72-76)
    seq = b"GCAGTATA" + b"CG" * 8 + b"TCATGATCGT" + b"AT" * 8 + b"GAGCACTC"
    go, ge, *_ = oracle.penalty_vectors(m, seq)
    assert go[12] == DI[8] - 2 and go[38] == DI[8]


def test_snv_model_masks_priors_and_substitution_mask():
    m = model()
    seq = b"GCAGTATCATGATCGTGAGCAC" + b"T" * 10 + b"CGTACGCAGTCTGAGTA"
    go, ge, mf, pf, mr, pr = oracle.penalty_vectors(m, seq)
    s = np.frombuffer(seq, np.uint8)
    assert np.array_equal(mf, np.roll(s, 1)) and np.array_equal(mr, np.roll(s, -1))   # masks = the haplotype rotated by one base (:173-177)
    assert pf[:15].max() == 125 and pf.min() < 125 and pr.min() < 125              # the homopolymer lowers the caps around it, on each strand's side
    assert int(np.argmin(pf)) >= 22 and int(np.argmin(pr)) <= 32
    sub = np.zeros(len(seq), np.uint8); sub[20:36] = 1
    *_, pf2, _, pr2 = oracle.penalty_vectors(m, seq, sub)
    assert set(pf2[20:36].tolist()) == {125} and set(pr2[20:36].tolist()) == {125}  # haplotype's own substitutions are never down-weighted (:168-172)
    assert np.array_equal(pf2[:20], pf[:20]) and np.array_equal(pf2[36:], pf[36:])


def ref_penalty_vectors(seq, substitution_mask=None):
    """The reference's own BasicRepeatBasedIndelErrorModel + BasicRepeatBasedSNVErrorModel (oracle/_ref) on the default tables."""
    import ctypes as C
    tabs = [AT, CG, DI, TRI, [3, 3, 3, 3, 3, 3, 4, 5, 6, 6, 8, 8, 7, 6, 5, 4, 3], [3, 3, 5, 4, 3, 2], [3, 3, 5, 4, 3, 2]] + SNV
    flat = np.asarray([v for t in tabs for v in t], np.int8); lens = np.asarray([len(t) for t in tabs], np.uint32)
    n = len(seq)
    go, ge, pf, pr = (np.zeros(n, np.int8) for _ in range(4)); mf, mr = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    p = lambda a: None if a is None else np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    sub = None if substitution_mask is None else np.ascontiguousarray(substitution_mask, dtype=np.uint8)
    oracle.ref().ref_error_models(p(flat), p(lens), bytes(seq), n, p(sub), p(go), p(ge), p(mf), p(pf), p(mr), p(pr))
    return go, ge, mf, pf, mr, pr


@pytest.mark.skipif(not oracle.have_ref(), reason="reference build absent")
def test_error_model_restatement_equals_the_reference_classes():
    """All six vectors of HaplotypeLikelihoodModel::reset: the restatement vs the reference's own model classes compiled in place
    (default PCR-free tables), on random haplotypes with planted repeats and random substitution masks."""
    rng = np.random.default_rng(77)
    m = model()
    names = ("gap_open", "gap_extend", "mask_fwd", "prior_fwd", "mask_rev", "prior_rev")
    for it in range(600):
        seq = random_sequence(rng, int(rng.integers(2, 420)), b"ACGT" if it % 6 else b"ACGTN")
        sub = None
        if it % 3 == 0:
            sub = np.zeros(len(seq), np.uint8)
            for _ in range(int(rng.integers(0, 4))):
                a = int(rng.integers(0, len(seq))); sub[a:a + int(rng.integers(1, 6))] = 1
        got, want = oracle.penalty_vectors(m, seq, sub), ref_penalty_vectors(seq, sub)
        for name, g, w in zip(names, got, want):
            assert np.array_equal(g, w), (name, seq, np.flatnonzero(g != w)[:5])


@pytest.mark.skipif(not oracle.have_ref(), reason="reference build absent")
def test_sort_by_length_restatement_equals_std_sort():
    """gap_extend depends on the order std::sort (unstable) leaves equal-length repeats in: the restated libstdc++ introsort against the real one,
    including inputs that reach its heap-sort fallback."""
    import ctypes as C
    rng = np.random.default_rng(0)
    def run(lib, fn, a):
        out = np.zeros(len(a), np.uint32)
        getattr(lib, fn)(a.ctypes.data_as(C.c_void_p), len(a), out.ctypes.data_as(C.c_void_p))
        return out
    cases = []
    for it in range(1500):
        a = rng.integers(2, 2 + int(rng.choice([2, 3, 5, 20, 1000])), int(rng.integers(0, 300))).astype(np.uint32)
        cases.append(np.sort(a) if it % 7 == 0 else (np.sort(a)[::-1].copy() if it % 11 == 0 else a))
    for n in (64, 128, 500, 1000, 4000):                                   # alternating low / high halves: deep recursion
        k = n // 2; a = np.zeros(n, np.uint32)
        a[0::2] = [i + 1 if i % 2 == 0 else k + i + 1 for i in range(k)]
        a[1::2] = [k + i + 1 if i % 2 == 0 else i + 1 for i in range(k)]
        cases.append(a)
    for a in cases:
        assert np.array_equal(run(oracle.ref(), "ref_sort_by_length", a), run(oracle.lib(), "oracle_sort_by_length", a))
