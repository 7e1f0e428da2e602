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
