"""The oracle's k-mer mapper restatement against the reference's own header (utils/kmer_mapper.hpp compiled in place into oracle/_ref):
same candidate offsets in the same order, including ties on repeat-rich haplotypes, non-ACGT bytes and tiny position caps."""
import ctypes as C

import numpy as np
import pytest

import oracle


@pytest.mark.skipif(not oracle.have_ref(), reason="reference build absent")
def test_kmer_mapper_restatement_equals_the_reference_header():
    R = oracle.ref()
    rng = np.random.default_rng(3)
    out = (C.c_uint32 * 64)()
    for it in range(2500):
        Lh, T = int(rng.integers(6, 400)), int(rng.integers(1, 160))
        alpha = b"ACGT" if it % 4 else b"ACGTN"
        t = bytes(alpha[i] for i in rng.integers(0, len(alpha), Lh))
        if rng.random() < 0.3:
            t = (t[:50] * 8)[:Lh]                                   # tandem copies: several offsets tie for the maximum
        if rng.random() < 0.6 and Lh > T:
            a = int(rng.integers(0, Lh - T + 1)); q = bytearray(t[a:a + T])
            for _ in range(int(rng.integers(0, 4))):
                q[int(rng.integers(0, T))] = alpha[int(rng.integers(0, len(alpha)))]
            q = bytes(q)
        else:
            q = bytes(alpha[i] for i in rng.integers(0, len(alpha), T))
        mp = int(rng.choice([1, 2, 5, 10, 10, 10, 15]))
        n = R.ref_kmer_map(q, len(q), t, len(t), mp, out)
        assert oracle.map_query_to_target(q, t, mp) == [out[i] for i in range(n)], (q, t, mp)
