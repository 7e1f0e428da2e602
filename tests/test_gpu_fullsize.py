"""Full-size parity on the BASELINE.json configs the headline numbers are quoted on (configs[2], [3], [4]): the GPU output of the very
batches bench.py times, with the product's defaults (slice pipeline, late traceback start), against the REFERENCE's own
HaplotypeLikelihoodArray::populate running on all host threads. See tests/check_fullsize.py."""
import pytest

import check_fullsize as cf
import oracle

pytestmark = pytest.mark.gpu


def need_ref(test):
    """These are the tests that carry full-size parity: on the GPU box the reference build (oracle/_ref, shipped with the snapshot) must be there - a box
    without it FAILS them instead of quietly skipping the three most important checks."""
    import functools
    from backends import require_reference_build

    @functools.wraps(test)
    def run(*a, **kw):
        require_reference_build(oracle.have_ref_array(), "oracle/_ref/libref_array.so")
        return test(*a, **kw)
    return run


@need_ref
def test_gpu_bench_batch_100k_by_128_equals_the_reference_populate():
    """BASELINE.json configs[2] = bench.py's default workload: synth.config_batch("100kx128", seed=42, B=16, positions="none"),
    12.8 M log-likelihoods, ~17.5 M DP tasks; 8 slices, late traceback start on (both engage from 100 k pairs)."""
    r = cf.check_bench_batch("gpu", "100kx128", B=16, seed=42)
    s = r["stats"]
    assert s["n_pairs"] == 12_800_000 and s["n_dp_traceback"] > 5_000_000 and s["n_dp_score_only"] > 2_000_000 and s["n_fast_path"] > 100_000
    print(f"100k x 128: {r['n']} values equal; reference populate {r['reference_s']:.1f} s = {r['reference_gcups']:.1f} GCUPS on {oracle.host_cores()} threads")


@need_ref
def test_gpu_region_stream_of_2000_regions_equals_the_reference_region_by_region():
    """BASELINE.json configs[3] stand-in = `bench.py --workload stream`: 2,000 synthetic active regions in one flat batch."""
    r = cf.check_region_stream("gpu", n_regions=2000, B=16, seed=42)
    assert r["regions"] == 2000 and r["stats"]["n_pairs"] > 10_000_000


@need_ref
def test_gpu_long_read_config_full_size_populate_and_align():
    """BASELINE.json configs[4] at full size: 64 x 10 kb reads, 8 x 20 kb haplotypes, band 256, int32 lanes."""
    r = cf.check_long_reads("gpu")
    assert r["n"] == 512 and r["n_alignments"] == 512 and r["stats"]["band_cells"] > 4_000_000_000
    assert r["stats"]["n_dp_traceback"] > 0 and r["stats"]["n_dp_score_only"] > 0


@need_ref
def test_gpu_linked_chunk_stream_equals_the_reference_region_by_region():
    """The reference's own long-read configuration (PacBioCCS.config: 500-base linked chunks, band 16): 200 regions, templates of 2-3 chunks."""
    r = cf.check_linked_stream("gpu", n_regions=200)
    assert r["regions"] == 200 and r["stats"]["n_pairs"] > 300_000 and r["stats"]["n_dp_traceback"] > 0
