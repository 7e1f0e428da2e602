"""Pin the oracle's populate driver (rows x haplotypes loop, per-batch read hashes, per-haplotype k-mer table and model reset, template sums,
sample concatenation, ShortHaplotypeError) on the REFERENCE's own HaplotypeLikelihoodArray: core/models/haplotype_likelihood_array.cpp
compiled in place into oracle/_ref/libref_array.so on stand-in read / haplotype / container types (oracle/ref_array_bridge.cpp).
Every comparison is exact equality of doubles. Skipped where /root/reference was never present (the prebuilt library travels to the GPU box)."""
import numpy as np
import pytest

import oracle
from octopus_amd import abi, synth

pytestmark = pytest.mark.skipif(not oracle.have_ref_array(), reason="oracle/_ref/libref_array.so not built (no /root/reference)")


def one_region(seed, R=30, H=5, T=60, Lh=160, B=8, flank=(20, 20), indels=0, with_n=False):
    rng = np.random.default_rng(seed)
    g = synth.make_region(rng, R, H, T=T, Lh=Lh, B=B, flank=flank, positions="none", indels_per_read=indels)
    if with_n:
        g["reads"][rng.integers(0, R, 3), rng.integers(0, T, 3)] = ord("N")
        g["haps"][1][int(rng.integers(0, Lh))] = ord("N")
    return synth.batch_from_regions([g]), rng


def both(batch, sample_rows=None, n_threads=1, merged=False, **cfg_kw):
    cfg = abi.Config.default(**cfg_kw)
    want, st, _ = oracle.populate(cfg, batch, n_threads=2)
    code, got, mg, err_hap, ext = oracle.ref_array_populate(cfg, batch, sample_rows=sample_rows, n_threads=n_threads, merged=merged)
    return want, st, code, got, mg, err_hap, ext


@pytest.mark.parametrize("B,flank,with_n", [(8, (20, 20), False), (16, (30, 25), False), (8, None, False), (8, (10, 40), True), (32, (30, 30), False)])
def test_read_map_overload_equals_the_oracle(B, flank, with_n):
    batch, _ = one_region(100 + B, T=50 + B, Lh=170 + 4 * B, B=B, flank=flank, indels=1, with_n=with_n)
    want, st, code, got, _, _, _ = both(batch, max_indel_error=B)
    assert st.code == abi.OK and code == 0
    assert np.array_equal(want, got)
    assert np.all(np.isfinite(got)) and np.any(got < 0)


@pytest.mark.parametrize("n_threads", [1, 4])
def test_template_map_overload_sums_reads_of_a_template_like_the_oracle(n_threads):
    """n_threads = 4 takes the reference's thread-pool branch (one task per haplotype, model copied per task, :167-184)."""
    batch, rng = one_region(7, R=41, H=6)
    rows, r = [0], 0
    while r < 41:
        r += min(41 - r, int(rng.integers(1, 4)))
        rows.append(r)
    batch.row_offsets = np.asarray(rows, np.uint32)
    want, st, code, got, _, _, _ = both(batch, n_threads=n_threads, max_indel_error=8)
    assert st.code == abi.OK and code == 0
    assert want.size == 6 * (len(rows) - 1)
    assert np.array_equal(want, got)


def test_samples_are_concatenated_rows_and_merge_samples_agrees():
    batch, _ = one_region(9, R=37, H=4)
    want, st, code, got, mg, _, _ = both(batch, sample_rows=[0, 11, 11, 30, 37], merged=True, max_indel_error=8)   # one sample has no reads
    assert st.code == abi.OK and code == 0
    assert np.array_equal(want, got)
    assert np.array_equal(want, mg)             # merge_samples(): the C ABI's flat rows are exactly the merged array


def test_mapping_quality_and_flank_options_reach_the_model():
    batch, rng = one_region(12, R=25, H=3)
    batch.mapq[:] = rng.integers(0, 70, len(batch.mapq))
    for kw in (dict(use_mapping_quality=0), dict(mapping_quality_cap=30), dict(mapping_quality_cap=40, mapping_quality_cap_trigger=20),
               dict(use_flank_state=0)):
        want, st, code, got, _, _, _ = both(batch, max_indel_error=8, **kw)
        assert st.code == abi.OK and code == 0
        assert np.array_equal(want, got), kw


def test_short_haplotype_error_names_the_same_haplotype_and_extension():
    rng = np.random.default_rng(3)
    B, Lh = 8, 100
    hap = synth.BASES[rng.integers(0, 4, Lh)]
    T = Lh - 2 * B + 5
    reads = [dict(seq=bytes(hap[10:60]), quals=np.full(50, 30, np.uint8), begin=10),
             dict(seq=bytes(hap[:T]), quals=np.full(T, 30, np.uint8), begin=1)]
    hl = []
    for h in (hap, hap.copy()):
        go, ge, mf, pf, mr, pr = synth._penalties(h)
        hl.append(dict(seq=bytes(h), begin=0, gap_open=go, gap_extend=ge, mask_fwd=mf, prior_fwd=pf, mask_rev=mr, prior_rev=pr))
    hl[1]["seq"] = bytes(hl[1]["seq"][:-1]) + (b"A" if hl[1]["seq"][-1:] != b"A" else b"C")
    batch = abi.Batch.from_lists(reads, hl, flank=(10, 10))
    want, st, code, got, _, err_hap, ext = both(batch, max_indel_error=B)
    assert st.code == abi.ESHORT_HAPLOTYPE and code == 1
    assert (err_hap, ext) == (st.hap_index, st.required_extension)


def test_random_regions_ragged_reads_templates_and_samples():
    rng = np.random.default_rng(2718)
    n_short = 0
    for it in range(40):
        B = int(rng.choice([8, 16]))
        T, Lh = int(rng.integers(30, 90)), int(rng.integers(150, 260))
        R, H = int(rng.integers(5, 30)), int(rng.integers(1, 6))
        flank = None if rng.random() < 0.25 else (int(rng.integers(0, 60)), int(rng.integers(0, 60)))
        g = synth.make_region(rng, R, H, T=T, Lh=Lh, B=B, flank=flank, positions="none", indels_per_read=int(rng.integers(0, 2)))
        if rng.random() < 0.3:
            g["reads"][rng.integers(0, R, 2), rng.integers(0, T, 2)] = ord("N")
        rl, hl = [], []
        for r in range(R):
            n = int(rng.integers(max(7, T // 2), T + 1))                                  # ragged; some reads shorter than the band
            begin = int(g["begin"][r]) + (int(rng.integers(-30, 30)) if rng.random() < 0.15 else 0)   # a few reads hang off an end
            rl.append(dict(seq=bytes(g["reads"][r][:n]), quals=g["quals"][r][:n], mapq=int(rng.integers(0, 70)), reverse=bool(g["reverse"][r]), begin=max(begin, 0)))
        for h in g["haps"]:
            go, ge, mf, pf, mr, pr = synth._penalties(h)
            hl.append(dict(seq=bytes(h), begin=0, gap_open=go, gap_extend=ge, mask_fwd=mf, prior_fwd=pf, mask_rev=mr, prior_rev=pr))
        templates = None
        if rng.random() < 0.5:
            templates, r = [], 0
            while r < R:
                k = min(R - r, int(rng.integers(1, 4)))
                templates.append(list(range(r, r + k))); r += k
        batch = abi.Batch.from_lists(rl, hl, flank=flank, templates=templates)
        n_rows = R if templates is None else len(templates)
        cuts = sorted(int(x) for x in rng.integers(0, n_rows + 1, int(rng.integers(0, 3))))
        want, st, code, got, mg, err_hap, ext = both(batch, sample_rows=[0] + cuts + [n_rows], merged=True, n_threads=int(rng.choice([1, 4])),
                                                     max_indel_error=B)
        if st.code == abi.ESHORT_HAPLOTYPE:
            n_short += 1
            assert code == 1 and (err_hap, ext) == (st.hap_index, st.required_extension), it
            continue
        assert st.code == abi.OK and code == 0, it
        assert np.array_equal(want, got), it
        assert np.array_equal(want, mg), it
    assert n_short < 20


def test_several_regions_in_one_call_equal_the_reference_called_once_per_region():
    """Several regions per call are the C ABI's own extension (the reference calls populate once per active region): the flat output must be
    the reference's per-region matrices one after the other."""
    rng = np.random.default_rng(99)
    shapes = [(12, 3, (20, 20)), (9, 4, None), (17, 2, (5, 45))]
    regions = [synth.make_region(rng, R, H, T=60, Lh=170, B=8, flank=fl, positions="none", indels_per_read=1) for R, H, fl in shapes]
    cfg = abi.Config.default(max_indel_error=8)
    want, st, _ = oracle.populate(cfg, synth.batch_from_regions(regions), n_threads=2)
    assert st.code == abi.OK
    parts = []
    for g in regions:
        code, got, _, _, _ = oracle.ref_array_populate(cfg, synth.batch_from_regions([g]))
        assert code == 0
        parts.append(got)
    assert np.array_equal(want, np.concatenate(parts))


@pytest.mark.skipif(not oracle.have_ref_array("avx2"), reason="no AVX2 build of the reference array (or no AVX2 CPU)")
def test_avx2_build_of_the_reference_array_gives_the_same_matrix():
    """bench.py's cpu_baseline times this build where the CPU allows (what -march=native makes of the reference for B = 16 x int16)."""
    batch, _ = one_region(77, R=40, H=5, T=100, Lh=260, B=16, flank=(40, 40), indels=1)
    cfg = abi.Config.default(max_indel_error=16)
    want, st, _ = oracle.populate(cfg, batch, n_threads=2)
    code, got, _, _, _ = oracle.ref_array_populate(cfg, batch, isa="avx2")
    assert st.code == abi.OK and code == 0 and np.array_equal(want, got)

