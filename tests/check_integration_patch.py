"""INTEGRATION.md's patch, compiled: the reference's OWN HaplotypeLikelihoodArray class (haplotype_likelihood_array.hpp:65-134 on the
oracle/ref_shim stand-in containers) with the bodies of its two populate() overloads (cpp:51-103, :105-199) replaced by
pack -> oct_phmm_populate -> scatter (integration/populate_on_device.inc, spliced in by oracle/make_patched_tree.py), against the
same class unpatched — through the class's own read-back methods: operator()(sample, Haplotype / IndexedHaplotype), extract_sample,
prime + operator[], num_likelihoods, merge_samples (both overloads), reset(kept haplotypes), contains, clear, and the ShortHaplotypeError
it throws (same haplotype, same required extension)."""
import numpy as np

import oracle
from octopus_amd import abi, synth


def scenario(rng, B, R, H, T, Lh, flank, templates, n_samples):
    g = synth.make_region(rng, R, H, T=T, Lh=Lh, B=B, flank=flank, positions="none", indels_per_read=1)
    g["mapq"] = rng.integers(0, 70, R).astype(np.uint8)
    batch = synth.batch_from_regions([g])
    n_rows = R
    if templates:
        rows, r = [0], 0
        while r < R:
            r += min(R - r, int(rng.integers(1, 4)))
            rows.append(r)
        batch.row_offsets = np.asarray(rows, np.uint32)
        n_rows = len(rows) - 1
    cuts = np.sort(rng.choice(np.arange(1, n_rows), size=n_samples - 1, replace=False)) if n_samples > 1 else []
    sample_rows = np.concatenate([[0], cuts, [n_rows]]).astype(np.uint32)
    return batch, sample_rows


def check(backend, tol=0.0):
    lib = "patched_" + backend
    rng = np.random.default_rng(2024)
    n = 0
    for B, R, H, T, Lh, flank, tmpl, ns, threads in ((8, 30, 5, 60, 170, (20, 25), False, 3, 1), (16, 40, 6, 100, 260, (40, 40), True, 2, 4),
                                                     (16, 25, 4, 150, 300, None, False, 1, 1), (32, 20, 3, 120, 330, (30, 60), True, 3, 4)):
        batch, sample_rows = scenario(rng, B, R, H, T, Lh, flank, tmpl, ns)
        cfg = abi.Config.default(max_indel_error=B, mapping_quality_cap=40, mapping_quality_cap_trigger=30) if B == 16 else abi.Config.default(max_indel_error=B)
        keep = np.sort(rng.choice(H, size=max(1, H // 2), replace=False))
        code0, want, wf, _, _ = oracle.ref_array_exercise(cfg, batch, sample_rows, keep, threads, lib="sse2")
        code1, got, gf, _, _ = oracle.ref_array_exercise(cfg, batch, sample_rows, keep, threads, lib=lib)
        assert code0 == 0 and code1 == 0, (code0, code1)
        assert np.array_equal(wf, gf), (wf, gf)
        used = ~np.isnan(want)
        assert np.array_equal(used, ~np.isnan(got))
        assert np.max(np.abs(want[used] - got[used])) <= tol, (B, tmpl)
        assert used[:4].all() and used[5, :len(keep)].all()
        n += int(used.sum())
    # ShortHaplotypeError: the patched class throws the reference's exception type naming the same haplotype with the same extension
    short = synth.make_region(rng, 20, 3, T=60, Lh=150, B=16, flank=(10, 10), positions="none")
    short["haps"] = [h[:80] for h in short["haps"]]
    short["begin"] = np.zeros_like(short["begin"])
    batch = synth.batch_from_regions([short])
    cfg = abi.Config.default(max_indel_error=16)
    rows = np.asarray([0, 20], np.uint32)
    a = oracle.ref_array_exercise(cfg, batch, rows, [0], 1, lib="sse2")
    b = oracle.ref_array_exercise(cfg, batch, rows, [0], 1, lib=lib)
    assert a[0] == 1 and b[0] == 1 and a[3:] == b[3:], (a[0], b[0], a[3:], b[3:])
    n += check_unsupported_regions_fall_back(lib)
    return n


def check_unsupported_regions_fall_back(lib):
    """A region the device path refuses (OCT_PHMM_EUNSUPPORTED: a haplotype of 65,536 bases or more with device k-mer mapping; until round 6 also reads from 32 k bases on) is answered by the reference's own body, kept in the patched class as
    populate_on_host: the patched class answers every call the unpatched one answers - same matrix, same read-backs; and the next region runs on the device again."""
    rng = np.random.default_rng(77)
    n = 0
    for tmpl in (False, True):
        g = synth.make_region(rng, 4, 2, T=400, Lh=65_900, B=8, flank=(30, 30), positions="none")
        batch = synth.batch_from_regions([g])
        cfg = abi.Config.default(max_indel_error=8)
        from backends import make_engine
        eng = make_engine(lib.split("_")[1], max_indel_error=8)                 # the product itself refuses this batch: that is what the patch falls back from
        _, st = eng.populate(batch, raise_on_error=False)
        eng.close()
        assert st.code == abi.EUNSUPPORTED, st.code
        rows = np.asarray([0, 2, 4], np.uint32) if not tmpl else np.asarray([0, 1, 2], np.uint32)
        if tmpl:
            batch.row_offsets = np.asarray([0, 2, 4], np.uint32)
        a = oracle.ref_array_exercise(cfg, batch, rows, [1], 1, lib="sse2")
        b = oracle.ref_array_exercise(cfg, batch, rows, [1], 1, lib=lib)
        assert a[0] == 0 and b[0] == 0, (a[0], b[0])
        used = ~np.isnan(a[1])
        assert np.array_equal(used, ~np.isnan(b[1])) and np.array_equal(a[1][used], b[1][used]) and np.array_equal(a[2], b[2])     # the SAME code ran: equal to the last bit
        n += int(used.sum())
    return n
