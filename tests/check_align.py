"""Backend-independent parity checks of oct_phmm_align (HaplotypeLikelihoodModel::align for every pair) against the oracle:
mapping position, CIGAR and likelihood (identical integer penalty; fp64 mixture within 1e-9)."""
import numpy as np
import pytest

import oracle
from backends import make_engine
from octopus_amd import abi, synth
from check_populate import mapper_positions

TOL = 1e-9


def compare_align(backend, batch, max_cigar_ops=64, n_threads=2, **cfg_kw):
    cfg = abi.Config.default(**cfg_kw)
    want, wst = oracle.align_batch(cfg, batch, max_cigar_ops, n_threads=n_threads)
    eng = make_engine(backend, **cfg_kw)
    got, st = eng.align(batch, max_cigar_ops, raise_on_error=False)
    eng.close()
    assert st.code == wst.code, (st.code, wst.code, st.message, wst.message)
    if st.code == abi.ESHORT_HAPLOTYPE:
        assert (st.hap_index, st.read_index, st.required_extension) == (wst.hap_index, wst.read_index, wst.required_extension)
        return None
    if st.code == abi.EINVAL:
        assert st.required_extension == wst.required_extension
        return None
    assert st.code == abi.OK
    bad = np.flatnonzero(got["mapping_position"] != want["mapping_position"])
    assert bad.size == 0, (bad[:5], got["mapping_position"][bad[:5]], want["mapping_position"][bad[:5]])
    diff = [e for e in range(len(want["cigar_strings"])) if got["cigar_strings"][e] != want["cigar_strings"][e]]
    assert not diff, (diff[:5], [got["cigar_strings"][e] for e in diff[:5]], [want["cigar_strings"][e] for e in diff[:5]])
    assert np.max(np.abs(got["likelihood"] - want["likelihood"]), initial=0.0) <= TOL
    return got


def check_align_basic(backend):
    rng = np.random.default_rng(31)
    # reads with sequencing errors and a few indel reads against SNV / indel haplotypes; flank state on; device k-mer mapping
    batch = synth.batch_from_regions([synth.make_region(rng, 80, 6, B=16, positions="none", indels_per_read=0),
                                      synth.make_region(rng, 50, 5, B=16, positions="none", indels_per_read=1)])
    got = compare_align(backend, batch, max_indel_error=16)
    ops = "".join(got["cigar_strings"])
    assert "I" in ops and "D" in ops and "X" in ops and "=" in ops
    assert any(c == "150=" for c in got["cigar_strings"])            # exact matches take the try_naive_align route
    # every CIGAR consumes the whole read
    for c, n in zip(got["cigar_strings"], got["n_cigar_ops"]):
        assert n > 0
        import re
        assert sum(int(l) for l, o in re.findall(r"(\d+)([=XID])", c) if o in "=XI") == 150
    return got


def check_align_positions_and_options(backend):
    rng = np.random.default_rng(32)
    g = synth.make_region(rng, 60, 5, B=8, positions="none", indels_per_read=1)
    batch = synth.batch_from_regions([g])
    mapper_positions(batch, rng=rng, junk=0.4)                       # host-provided candidates in arbitrary order, duplicates, out-of-range ones
    compare_align(backend, batch, max_indel_error=8)
    compare_align(backend, batch, max_indel_error=8, use_flank_state=0)
    compare_align(backend, batch, max_indel_error=8, use_mapping_quality=0)
    compare_align(backend, batch, max_indel_error=8, mapping_quality_cap=30, mapping_quality_cap_trigger=20)
    # int32 lanes and a wide band (streaming DP kernels + their walkers)
    compare_align(backend, synth.batch_from_regions([synth.make_region(rng, 30, 4, B=16, positions="none", indels_per_read=1)]),
                  max_indel_error=16, use_int_scores=1)
    compare_align(backend, synth.batch_from_regions([synth.make_region(rng, 12, 3, T=400, Lh=1000, B=128, flank=(150, 150), positions="none",
                                                                       indels_per_read=3)]), max_indel_error=128)


def check_align_errors(backend):
    rng = np.random.default_rng(33)
    g = synth.make_region(rng, 40, 4, B=16, positions="none", indels_per_read=2)
    batch = synth.batch_from_regions([g])
    compare_align(backend, batch, max_cigar_ops=2, max_indel_error=16)          # capacity too small -> EINVAL + needed operations
    short = synth.make_region(rng, 20, 3, B=16, positions="none")
    short["haps"] = [h[:170] for h in short["haps"]]                             # haplotypes shorter than read + 2 * band
    short["begin"] = np.zeros_like(short["begin"])
    compare_align(backend, synth.batch_from_regions([short]), max_indel_error=16)
    b2 = synth.batch_from_regions([synth.make_region(rng, 10, 2, B=16, positions="none")])
    b2.row_offsets = np.arange(0, 11, 2, dtype=np.uint32)                        # templates are not alignable
    eng = make_engine(backend, max_indel_error=16)
    with pytest.raises(Exception) as e:
        eng.align(b2)
    assert e.value.code == abi.EINVAL
    eng.close()


def check_align_candidate_counts(backend):
    """oct_phmm_align_candidate_counts: reads that lie entirely inside a (CA)n run longer than themselves tie on more diagonals than any cap of the ABI keeps (the reference's
    realigner maps without a cap, read_realigner.cpp:128,137): their pairs report cap candidates and are counted as saturated; ordinary reads report what the oracle's mapper finds.
    With caller-provided positions nothing is saturated."""
    rng = np.random.default_rng(61)
    T, Lh, B = 60, 420, 16
    g = synth.make_region(rng, 12, 2, T=T, Lh=Lh, B=B, flank=None, positions="none")
    for h in g["haps"]:
        h[150:290] = np.frombuffer(b"CA" * 70, np.uint8)
        h[149] = ord("T"); h[290] = ord("G")
    inside = [0, 3, 7]
    for k, r in enumerate(inside):
        o = 150 + 2 * (5 + 7 * k)
        g["reads"][r] = g["haps"][0][o:o + T]; g["begin"][r] = o
    batch = synth.batch_from_regions([g])
    for cap in (15, 4):
        cfg = abi.Config.default(max_indel_error=B, max_mapping_positions=cap)
        eng = make_engine(backend, max_indel_error=B, max_mapping_positions=cap)
        got, st = eng.align(batch, 96)
        assert st.code == abi.OK
        counts, n_sat = eng.align_candidate_counts(batch.n_read_pairs())
        want = []
        for (h, r) in batch.read_pairs():
            hs = bytes(batch.hap_bases[batch.hap_offsets[h]:batch.hap_offsets[h + 1]]); rs = bytes(batch.read_bases[batch.read_offsets[r]:batch.read_offsets[r + 1]])
            want.append(min(cap, len(oracle.map_query_to_target(rs, hs, 1000))))
        assert counts.tolist() == want, (cap, counts.tolist(), want)
        assert n_sat == sum(1 for w in want if w >= cap) and n_sat >= 2 * len(inside)
        with pytest.raises(Exception):
            eng.align_candidate_counts(batch.n_read_pairs() + 1)      # not the last call's pair count
        # caller-provided positions: the library cut nothing short
        b2 = mapper_positions(synth.batch_from_regions([g]), max_positions=cap, rng=None)
        eng.align(b2, 96)
        assert eng.align_candidate_counts(b2.n_read_pairs())[1] == 0
        eng.close()
    return True


def check_align_reads_beyond_32k_bases(backend, T=33_300, Lh=34_132):
    """oct_phmm_align on reads of 32,768 bases and more (refused until round 6, see check_populate.check_reads_beyond_32k_bases): position, CIGAR and likelihood of every pair
    against the oracle - band 16, int32 lanes (the reference's long-read configuration), device k-mer mapping."""
    import numpy as np
    from octopus_amd import synth
    rng = np.random.default_rng(95)
    g = synth.make_region(rng, 3, 2, T=T, Lh=Lh, B=16, flank=(300, 300), positions="none", indels_per_read=6)
    g["quals"][:] = np.clip(g["quals"], 5, 25)
    return compare_align(backend, synth.batch_from_regions([g]), max_cigar_ops=4096, max_indel_error=16, use_int_scores=1)

