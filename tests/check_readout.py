"""Backend-independent checks of the genotype read-out (oct_phmm_batch_genotype_likelihoods) against the oracle's restatement of
ConstantMixtureGenotypeLikelihoodModel::evaluate. The matrix both sides read is the one the backend's own populate produced
(its parity with the oracle is checked in check_populate.py), so any difference here is the read-out's."""
import itertools

import numpy as np
import pytest

import oracle
from backends import make_engine
from octopus_amd import abi, synth

RTOL = 1e-11      # fp64 sums of up to a few thousand terms in a different association + libm vs device exp/log: far inside the 1e-4 bar


def all_genotypes(haps, ploidy):
    return np.asarray(list(itertools.combinations_with_replacement(haps, ploidy)), np.uint32)


def random_genotypes(rng, haps, ploidy, n):
    g = np.sort(rng.choice(np.asarray(haps), size=(n, ploidy), replace=True), axis=1)
    return g.astype(np.uint32)


def zygosity_patterns(haps, ploidy):
    """Every way `ploidy` sorted slots can repeat (e.g. aabc, abbc, abcc, aabb ... for 4) on the first few haplotypes."""
    out = set()
    for combo in itertools.product(range(ploidy), repeat=ploidy):
        if list(combo) == sorted(combo) and combo[0] == 0 and all(b - a <= 1 for a, b in zip(combo, combo[1:])):
            out.add(tuple(haps[c] for c in combo))
    return np.asarray(sorted(out), np.uint32)


def _close(got, want):
    bad = np.flatnonzero(~(np.abs(got - want) <= RTOL * np.maximum(1.0, np.abs(want))))
    assert bad.size == 0, (bad[:5], got[bad[:5]], want[bad[:5]])


def check_readout(backend, seed=5, n_regions=3, big=False):
    rng = np.random.default_rng(seed)
    if big:
        regions = [synth.make_region(rng, 6000, 40, B=16, positions="true")]
    else:
        regions = [synth.make_region(rng, int(rng.integers(30, 90)), int(rng.integers(3, 9)), B=16, positions="true") for _ in range(n_regions)]
    batch = synth.batch_from_regions(regions)
    eng = make_engine(backend, max_indel_error=16)
    rb = eng.upload(batch)
    rb.run()
    lik = rb.download()
    off = batch.hap_out_offsets()
    ro, ho = batch.region_tables()
    sets = []
    for g in range(len(ro) - 1):
        haps = list(range(int(ho[g]), int(ho[g + 1])))
        rows = int(ro[g + 1] - ro[g])
        sub = (rows // 4, rows - rows // 5)                     # a sample's slice of the rows
        sets.append(dict(genotypes=all_genotypes(haps, 1), rows=(0, rows)))
        sets.append(dict(genotypes=all_genotypes(haps, 2), rows=(0, rows)))
        sets.append(dict(genotypes=all_genotypes(haps, 2), rows=sub))
        for p in (3, 4):
            pats = zygosity_patterns(haps, p) if len(haps) >= p else np.zeros((0, p), np.uint32)
            sets.append(dict(genotypes=np.concatenate([pats, random_genotypes(rng, haps, p, 300 if big else 40)]), rows=sub))
        for p in (5, 7, 16):
            sets.append(dict(genotypes=random_genotypes(rng, haps, p, 30), rows=(0, rows)))
        sets.append(dict(genotypes=all_genotypes(haps[:2], 2), rows=(rows // 2, rows // 2)))      # empty row range -> 0
    got = rb.genotype_likelihoods(sets)
    pos = 0
    for s in sets:
        n = len(s["genotypes"])
        want = oracle.genotype_likelihoods(lik, off, s["genotypes"], s["rows"])
        _close(got[pos:pos + n], want)
        if s["rows"][0] == s["rows"][1]:
            assert np.all(got[pos:pos + n] == 0.0)
        pos += n
    assert pos == len(got)
    # the haploid read-out is the plain column sum (haplotype_filter.cpp LikelihoodSum)
    h0 = int(ho[0]); rows0 = int(ro[1] - ro[0])
    one = rb.genotype_likelihoods([dict(genotypes=np.asarray([[h0]], np.uint32), rows=(0, rows0))])
    assert abs(one[0] - lik[int(off[h0]):int(off[h0]) + rows0].sum()) <= RTOL * max(1.0, abs(one[0]))
    # default row range = the whole region
    d = rb.genotype_likelihoods([dict(genotypes=all_genotypes(list(range(int(ho[0]), int(ho[1]))), 2))])
    _close(d, oracle.genotype_likelihoods(lik, off, all_genotypes(list(range(int(ho[0]), int(ho[1]))), 2), (0, rows0)))
    rb.free(); eng.close()


def check_readout_errors(backend):
    rng = np.random.default_rng(3)
    batch = synth.batch_from_regions([synth.make_region(rng, 30, 4, B=16, positions="true") for _ in range(2)])
    eng = make_engine(backend, max_indel_error=16)
    rb = eng.upload(batch)
    with pytest.raises(Exception) as e:          # not run yet
        rb.genotype_likelihoods([dict(genotypes=np.asarray([[0, 1]], np.uint32))])
    assert e.value.code == abi.EINVAL
    rb.run()
    for bad, code in (([[1, 0]], abi.EINVAL),                 # not sorted
                      ([[0, 5]], abi.EINVAL),                 # two regions in one genotype
                      ([[0, 99]], abi.EINVAL),                # no such haplotype
                      ([[0] * 17], abi.EUNSUPPORTED)):        # ploidy > 16
        with pytest.raises(Exception) as e:
            rb.genotype_likelihoods([dict(genotypes=np.asarray(bad, np.uint32))])
        assert e.value.code == code, (bad, e.value.code)
    with pytest.raises(Exception) as e:                        # rows outside the region
        rb.genotype_likelihoods([dict(genotypes=np.asarray([[0, 1]], np.uint32), rows=(0, 31))])
    assert e.value.code == abi.EINVAL
    assert len(rb.genotype_likelihoods([])) == 0
    rb.free(); eng.close()
