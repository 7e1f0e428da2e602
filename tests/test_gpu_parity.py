"""Parity tests proper: the HIP library on a real MI355X, through the C ABI, against the oracle and the
reference's golden vectors. Integer scores / alignments / flank scores must be identical; ln-likelihoods agree to
1e-9 (north-star tolerance is 1e-4; the only non-integer step is the fp64 mapping-quality mixture, where device
log/exp may differ from glibc in the last ulps)."""
import numpy as np
import pytest

import check_host_mirror
import check_l1
import check_populate as cp
import oracle
from backends import make_engine, require_reference_build
from octopus_amd import abi, synth

pytestmark = pytest.mark.gpu
TOL = 1e-9


def test_gpu_golden_vectors(golden_records):
    assert check_l1.check_golden("gpu", golden_records) == 22


@pytest.mark.parametrize("band,n", [(8, 300), (16, 600), (32, 200), (64, 100)])
def test_gpu_random_windows_fast_kernel(band, n):
    check_l1.check_random("gpu", band, n, seed=1000 + band, with_n=False)


@pytest.mark.parametrize("band,n", [(8, 200), (16, 300), (32, 100), (64, 50)])
def test_gpu_random_windows_generic_kernel(band, n):
    check_l1.check_random("gpu", band, n, seed=2000 + band, with_n=True)


def test_gpu_int32_lanes_golden_vectors(golden_records):
    assert check_l1.check_golden("gpu", golden_records, score_bits=32) == 17


@pytest.mark.parametrize("band,n", [(8, 100), (16, 200), (32, 60), (64, 30)])
def test_gpu_int32_lanes_random_windows(band, n):
    check_l1.check_random("gpu", band, n, seed=3000 + band, with_n=True, score_bits=32)


def test_gpu_populate_int32_lanes():
    cp.check_int32_lanes("gpu", TOL)


def test_gpu_exact_add_mode_gives_same_results(monkeypatch):
    monkeypatch.setenv("OCT_PHMM_EXACT_ADDS", "1")
    check_l1.check_random("gpu", 16, 200, seed=1016, with_n=False)
    check_l1.check_random("gpu", 8, 100, seed=2008, with_n=True)
    cp.check_basic("gpu", TOL)


@pytest.mark.parametrize("band,bits,n", [(128, 16, 40), (128, 32, 30), (256, 32, 20), (256, 16, 20)])
def test_gpu_wide_bands_streaming_kernel(band, bits, n):
    check_l1.check_random("gpu", band, n, seed=4000 + band + bits, t_lo=40, t_hi=300, with_n=True, score_bits=bits)


def test_gpu_populate_wide_bands_and_long_reads():
    """Includes a reduced BASELINE.json configs[4]: 3 kb reads x 12 kb haplotypes (big-haplotype k-mer mapper), band 256, int32 lanes, traceback spilled to HBM."""
    cp.check_wide_and_long("gpu", TOL, long_T=3000, long_Lh=12000, n_reads=6)


def test_gpu_unmasked_overload():
    check_l1.check_random("gpu", 16, 100, seed=7, masked=False, with_n=False)


def test_gpu_int16_overflow_wraps_like_reference():
    check_l1.check_random("gpu", 16, 100, seed=9, t_lo=150, t_hi=151, q_max=125, junk=True, with_n=False)


def test_gpu_populate_basic():
    cp.check_basic("gpu", TOL)


def test_gpu_populate_generic_bytes():
    cp.check_generic_bytes("gpu", TOL)


def test_gpu_populate_templates_and_regions():
    cp.check_templates_and_regions("gpu", TOL)


def test_gpu_populate_ragged_reads_edges_and_short_haplotype():
    cp.check_ragged_and_edges("gpu", TOL)


def test_gpu_populate_mapping_quality_and_flank_options():
    cp.check_mapping_quality_options("gpu", TOL)


def test_gpu_device_kmer_mapper_matches_reference_mapper():
    cp.check_device_kmer_mapper("gpu", TOL)


def test_gpu_device_mapper_positions_equal_the_oracle_mapper_on_tie_rich_haplotypes(monkeypatch):
    assert cp.check_kmer_mapper_positions("gpu") >= 200
    monkeypatch.setenv("OCT_PHMM_BIG_MAPPER", "1")
    assert cp.check_kmer_mapper_positions("gpu", seeds=(71, 72)) >= 50


def test_gpu_config2_with_device_mapping_matches_oracle():
    """configs[1] again, but with positions == NULL: 6-mer mapping runs on the device (oracle maps on the CPU)."""
    batch = synth.config_batch("1kx64", seed=42, B=16, positions="none")
    stats = cp.compare("gpu", batch, TOL, max_indel_error=16)
    assert stats["n_pairs"] == 64000


def test_gpu_populate_in_slices(monkeypatch):
    monkeypatch.setenv("OCT_PHMM_SLICES", "4")
    cp.check_basic("gpu", TOL)
    cp.check_templates_and_regions("gpu", TOL)
    cp.check_device_kmer_mapper("gpu", TOL)
    cp.check_ragged_and_edges("gpu", TOL)
    batch = synth.config_batch("1kx64", seed=43, B=16, positions="none")
    cp.compare("gpu", batch, TOL, max_indel_error=16)
    cp.check_one_shot_populate_streams_slices_back("gpu")


def test_gpu_config2_1k_by_64_matches_oracle():
    """BASELINE.json configs[1]: the 1k x 64 batch (150 bp reads, 300 bp haplotypes, B = 16, flank 40/40)."""
    batch = synth.config_batch("1kx64", seed=42, B=16)
    stats = cp.compare("gpu", batch, TOL, max_indel_error=16)
    assert stats["n_pairs"] == 64000 and stats["n_dp_traceback"] > 1000 and stats["n_fast_path"] > 1000


def test_gpu_cpp_host_mirror_matches_oracle_and_maps_errors():
    check_host_mirror.check("gpu", TOL)


def test_gpu_large_batch_properties():
    """At a size the oracle cannot check in seconds: size-independent properties. (a) every ln-likelihood <= 0 and
    finite, (b) idempotence: running the resident batch twice gives identical bytes, (c) a sub-batch of the same
    reads/haplotypes reproduces the corresponding rows exactly, and that sub-batch matches the oracle."""
    rng = np.random.default_rng(5)
    g = synth.make_region(rng, 20000, 32, B=16)
    batch = synth.batch_from_regions([g])
    eng = make_engine("gpu", max_indel_error=16)
    rb = eng.upload(batch)
    rb.run(); a = rb.download().copy()
    rb.run(); b = rb.download().copy()
    rb.free()
    assert np.array_equal(a, b)
    assert np.all(np.isfinite(a)) and np.all(a <= 0)
    m = a.reshape(32, 20000)
    sub = dict(g); sub["reads"] = g["reads"][:500]; sub["quals"] = g["quals"][:500]; sub["begin"] = g["begin"][:500]
    sub["reverse"] = g["reverse"][:500]; sub["mapq"] = g["mapq"][:500]; sub["pos"] = g["pos"][:, :500]
    small, _ = eng.populate(synth.batch_from_regions([sub]))
    assert np.array_equal(small.reshape(32, 500), m[:, :500])
    want, _, _ = oracle.populate(abi.Config.default(max_indel_error=16), synth.batch_from_regions([sub]), n_threads=4)
    assert np.max(np.abs(want - small)) <= TOL
    eng.close()


def test_gpu_genotype_readout_all_ploidies_and_zygosities():
    import check_readout as cr
    cr.check_readout("gpu")
    cr.check_readout_errors("gpu")


def test_gpu_genotype_readout_large_region_row_splits():
    """6,000 rows x 40 haplotypes: every genotype chunk is split over many row ranges and recombined in a fixed order."""
    import check_readout as cr
    cr.check_readout("gpu", seed=11, big=True)


def test_gpu_concurrent_handles_from_host_threads():
    """The reference calls populate from many region-task threads at once (caller.cpp:475, octopus.cpp:867): distinct handles must
    be usable concurrently from distinct host threads (ctypes releases the GIL during the call) and give the serial answers."""
    import threading
    n_threads, per_thread = 4, 6
    rng = np.random.default_rng(77)
    batches = [[synth.batch_from_regions([synth.make_region(rng, int(rng.integers(100, 600)), int(rng.integers(4, 20)), B=16, positions="none")])
                for _ in range(per_thread)] for _ in range(n_threads)]
    ref_eng = make_engine("gpu", max_indel_error=16)
    want = [[ref_eng.populate(b)[0].copy() for b in bs] for bs in batches]
    ref_eng.close()
    got = [[None] * per_thread for _ in range(n_threads)]
    errors = []

    def worker(t):
        try:
            eng = make_engine("gpu", max_indel_error=16)
            for rep in range(3):
                for i, b in enumerate(batches[t]):
                    got[t][i] = eng.populate(b)[0].copy()
            eng.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errors, errors
    for t in range(n_threads):
        for i in range(per_thread):
            assert np.array_equal(got[t][i], want[t][i])


def test_gpu_align_best_alignment_and_cigar_per_pair():
    import check_align as ca
    ca.check_align_basic("gpu")
    ca.check_align_positions_and_options("gpu")
    ca.check_align_errors("gpu")


def test_gpu_server_contract_violation_reaches_only_its_caller():
    import check_server
    assert check_server.check_server_contract_violation_reaches_only_its_caller("gpu")


def test_gpu_align_candidate_counts_name_the_saturated_pairs():
    import check_align
    assert check_align.check_align_candidate_counts("gpu")


def test_gpu_align_realigner_sized_batch():
    """read_realigner's shape: many reads against few haplotypes, device mapping; vs the oracle on all pairs."""
    import check_align as ca
    rng = np.random.default_rng(41)
    batch = synth.batch_from_regions([synth.make_region(rng, 3000, 4, B=16, positions="none")])
    got = ca.compare_align("gpu", batch, max_indel_error=16)
    assert len(got["cigar_strings"]) == 12000


def test_gpu_long_reads_at_narrow_bands_stream():
    stats = cp.check_long_reads_at_narrow_bands("gpu", TOL)
    assert all(s["n_dp_score_only"] + s["n_dp_traceback"] > 0 for s in stats) and any(s["n_dp_traceback"] > 0 for s in stats)
    # full PacBio-CCS-like shape: 10 kb reads, 14 kb haplotypes, band 16, int32 lanes
    import check_align as ca
    rng = np.random.default_rng(75)
    g = synth.make_region(rng, 12, 3, T=10_000, Lh=14_000, B=16, flank=(200, 200), positions="none", q_values=(20, 40), indels_per_read=3)
    batch = synth.batch_from_regions([g])
    cp.compare("gpu", batch, TOL, max_indel_error=16, use_int_scores=1)
    ca.compare_align("gpu", batch, max_cigar_ops=256, max_indel_error=16, use_int_scores=1)


def test_gpu_multi_region_shape_fuzz_slice():
    """A 100-scenario slice of the multi-region shape fuzz (tests/check_shapes.py; the round's full run is profiles/r05_final_gpu_fuzz.log): 1-64 regions, every
    haplotype count 1-48 and up to 400, R 20-5,000, ragged reads, linked chunks, NULL and given penalty vectors, random switches, through oct_phmm_populate, the
    resident API and the region server from 8 threads; sampled regions against the REFERENCE's own populate (oracle/_ref/libref_array.so must be on the box)."""
    import check_shapes
    require_reference_build(oracle.have_ref_array(), "oracle/_ref/libref_array.so")
    with check_shapes.worker_pool(10) as pool:
        st = check_shapes.check_shapes("gpu", range(5000, 5100), pool=pool)
    assert st["scenarios"] == 100 and st["sampled_regions"] >= 150 and st["server_calls"] > 100 and st["resident"] > 5 and st["null_vectors"] > 5, st


def test_gpu_random_scenarios():
    import check_fuzz
    assert check_fuzz.check_fuzz("gpu", seed=7, n=120, tol=TOL) == 120


def test_gpu_late_traceback_start_equals_oracle_and_plain_path(monkeypatch):
    cp.check_late_traceback_start("gpu", TOL)
    import check_fuzz
    monkeypatch.setenv("OCT_PHMM_LATE_MIN_PAIRS", "0")
    assert check_fuzz.check_fuzz("gpu", seed=31, n=80, tol=TOL) == 80
    batch = synth.config_batch("1kx64", seed=44, B=16, positions="none")
    cp.compare("gpu", batch, TOL, max_indel_error=16)


def test_gpu_reads_beyond_32k_bases():
    """Round 6: reads of 32,768 bases and more (refused until then: walk events held 15-bit coordinates) through k_walk_rows (band 16, int32), k_walk_long (band 256, int32) and
    the lockstep walker behind the streaming kernel (band 64, int16), device k-mer mapping, against the oracle; T + 2B >= 2^20 is still OCT_PHMM_EUNSUPPORTED."""
    assert len(cp.check_reads_beyond_32k_bases("gpu", TOL)) == 3
    import check_align as ca
    ca.check_align_reads_beyond_32k_bases("gpu")


def test_gpu_device_sized_and_host_sized_launches_agree(monkeypatch):
    """Region-sized batches take the device-sized launches (no mid-step read-back); the host-sized single-slice path on the same inputs."""
    assert cp.check_launch_modes("gpu", TOL) == 6
    import check_fuzz
    monkeypatch.setenv("OCT_PHMM_DEVICE_SIZED", "0")
    assert check_fuzz.check_fuzz("gpu", seed=77, n=60, tol=TOL) == 60
    cp.check_basic("gpu", TOL)
    cp.check_generic_bytes("gpu", TOL)


def test_gpu_staged_and_unstaged_walk_agree(monkeypatch):
    """Region-sized traceback launches give every walk a 16-lane row (2); big launches walk 64 tasks in lockstep with one line per lane in registers (0);
    the lockstep walker out of LDS-staged tiles (1) is the A/B form. All three on small inputs."""
    import check_fuzz
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("OCT_PHMM_WALK_STAGE", mode)
        cp.check_basic("gpu", TOL)
        cp.check_templates_and_regions("gpu", TOL)
        cp.check_late_traceback_start("gpu", TOL)
        cp.check_int32_lanes("gpu", TOL)
        assert check_fuzz.check_fuzz("gpu", seed=90 + int(mode), n=40, tol=TOL) == 40
    batch = synth.config_batch("1kx64", seed=45, B=16, positions="none")
    cp.compare("gpu", batch, TOL, max_indel_error=16)


@pytest.mark.parametrize("band,n", [(128, 60), (256, 60)])
def test_gpu_multi_wave_kernel_fast_cost_and_one_wave_form(band, n, monkeypatch):
    check_l1.check_random("gpu", band, n, seed=500 + band, t_lo=30, t_hi=700, with_n=False, score_bits=32)
    check_l1.check_random("gpu", band, n, seed=510 + band, t_lo=30, t_hi=700, with_n=True, score_bits=32)
    check_l1.check_random("gpu", band, n // 2, seed=520 + band, t_lo=200, t_hi=900, q_max=125, junk=True, with_n=False, score_bits=32)   # walks that wander over the band
    cp.check_wide_and_long("gpu", TOL)
    monkeypatch.setenv("OCT_PHMM_MW_PLANES", "1")         # all planes of a task in one wave (what a launch of >= 640 tasks takes)
    check_l1.check_random("gpu", band, n, seed=530 + band, t_lo=30, t_hi=700, with_n=False, score_bits=32)
    check_l1.check_random("gpu", band, n, seed=540 + band, t_lo=30, t_hi=700, with_n=True, score_bits=32)
    cp.check_wide_and_long("gpu", TOL)
    monkeypatch.delenv("OCT_PHMM_MW_PLANES")
    monkeypatch.setenv("OCT_PHMM_MULTI_WAVE", "0")
    monkeypatch.setenv("OCT_PHMM_WALK_STAGE", "0")        # ... and the lockstep walker instead of k_walk_long
    check_l1.check_random("gpu", band, n // 2, seed=600 + band, t_lo=40, t_hi=400, with_n=True, score_bits=32)
    cp.check_wide_and_long("gpu", TOL)


def test_gpu_scratch_allocation_failures_fall_back_and_leave_no_error():
    cp.check_scratch_allocation_failures("gpu", TOL)


def test_gpu_chunked_traceback_launches():
    cp.check_chunked_traceback("gpu", TOL)


def test_gpu_matrix_equals_the_reference_array_populate():
    """Device output against the reference's own HaplotypeLikelihoodArray::populate (prebuilt oracle/_ref/libref_array.so travels with the snapshot)."""
    require_reference_build(oracle.have_ref_array(), "oracle/_ref/libref_array.so")
    assert cp.check_against_reference_array("gpu", TOL) > 400


def test_gpu_streamed_upload_and_growing_calls():
    cp.check_streamed_upload_and_growing_calls("gpu", TOL)


def test_gpu_empty_batches():
    cp.check_empty_batches("gpu")


def test_gpu_matrix_equals_the_reference_composed_piece_by_piece():
    """The reference's own error models, k-mer mapper and likelihood model (oracle/_ref) produce the matrix; the GPU must reproduce it."""
    require_reference_build(oracle.have_ref(), "oracle/_ref/libref_phmm.so")
    from test_oracle_l3 import check_populate_composed_from_the_reference_pieces
    check_populate_composed_from_the_reference_pieces("gpu", TOL)


def test_gpu_server_over_several_devices_gives_the_same_answers():
    """oct_phmm_server_create_multi with every visible GPU (a one-GPU box lists its GPU twice: two sets of workers and handles)."""
    import check_server
    from octopus_amd import engine
    n = engine.load().oct_phmm_device_count()
    assert n >= 1
    devices = list(range(n)) if n > 1 else [0, 0]
    calls, batches = check_server.check_server("gpu", n_threads=8, per_thread=10, band=16, seed=29, devices=devices)
    assert calls == 160


def test_gpu_server_batches_concurrent_region_calls():
    import check_server
    calls, batches = check_server.check_server("gpu", n_threads=8, per_thread=12, band=16)
    assert calls == 192 and batches <= calls           # (whether two of these small calls ever queue up together depends on the box: a call takes 0.2 ms now;
                                                       #  the simulator's test pins the batching itself, tools/region_calls_bench measures it under load)


def test_gpu_server_answers_a_malformed_call_with_einval():
    import check_server
    check_server.check_server_rejects_malformed_calls("gpu")


def test_gpu_leading_slice_without_pairs_still_hashes_the_reads():
    cp.check_leading_haplotypes_without_reads("gpu", TOL)


def test_gpu_patched_reference_class_equals_the_unpatched_one():
    """INTEGRATION.md's patch compiled into the reference's own class, linked against liboct_phmm.so (prebuilt oracle/_ref/libref_array_patched_gpu.so)."""
    require_reference_build(oracle.have_ref_array() and oracle.have_patched_array("gpu"), "oracle/_ref/libref_array_patched_gpu.so")
    import check_integration_patch as ci
    assert ci.check("gpu", TOL) > 2000


def test_gpu_patched_read_assigner_seam_equals_the_reference_functions():
    """INTEGRATION.md's second seam (read_assigner.cpp:145-287) with its last function replaced by one oct_phmm_populate call, linked against
    liboct_phmm.so (prebuilt oracle/_ref/libref_assigner_patched_gpu.so), against the reference's own functions' committed output."""
    import check_assigner_patch as ca
    require_reference_build(ca.have("patched_gpu"), "oracle/_ref/libref_assigner_patched_gpu.so")
    # against the matrices the reference's functions gave where tests/golden/make_assigner_seam_golden.py ran (the CPU suite compares the two libraries directly)
    assert ca.check("gpu", TOL, golden=True) > 150


def test_gpu_pairs_with_equal_candidates_share_one_result():
    """Exact de-duplication of pairs: identical matrices with and without it, reference-equivalent counters unchanged, shared counters > 0."""
    assert len(cp.check_shared_pairs("gpu", TOL)) == 4


def test_gpu_populate_generates_the_penalty_vectors_on_host_threads_and_on_the_device():
    """SURVEY 8f-3 in the product: NULL vectors + oct_phmm_set_error_model; both generation paths equal the reference's error-model classes."""
    require_reference_build(oracle.have_ref(), "oracle/_ref/libref_phmm.so")
    import check_error_model as ce
    assert ce.check_populate_generates_the_vectors("gpu", TOL) > 0


def test_gpu_device_penalty_vectors_on_the_corpus_and_on_many_haplotypes():
    """k_penalty_vectors_wave (one haplotype per wave, workspace in LDS) and k_penalty_vectors (one haplotype per lane, workspace in HBM) on the
    2,400-string corpus of the error-model tests with substitution masks."""
    import check_error_model as ce
    assert ce.check_device_kernels_on_the_corpus("gpu", 2400) == 2400


def test_gpu_align_and_server_generate_the_penalty_vectors():
    import check_error_model as ce
    assert ce.check_align_and_server_generate_the_vectors("gpu", TOL) >= 5


def test_gpu_model_file_vectors_equal_the_reference_custom_model_and_feed_the_calls():
    """CustomRepeatBasedIndelErrorModel (`--sequence-error-model <file>`) through the product library: files accepted / refused like the reference's reader, vectors equal its
    class's; NULL-vector calls of a handle and of the region server run on them."""
    require_reference_build(oracle.have_ref(), "oracle/_ref/libref_phmm.so")
    import check_error_model as ce
    n_ok, n_bad = ce.check_custom_model_file(None)
    assert n_ok >= 80 and n_bad >= 50
    assert ce.check_custom_model_in_calls("gpu", TOL) > 0


def test_gpu_mapper_mismatch_account_feeds_the_fast_path():
    """k_kmer_map's account of the base mismatches along the mapped position (pair_mm) against the oracle and against the run without it."""
    assert len(cp.check_mapper_mismatch_account("gpu", TOL)) == 5


def test_gpu_haplotypes_of_40k_bases_and_more_map_on_the_device():
    """k_kmer_map_big with 16-bit counters: a 60 k-base haplotype (refused until round 5), 64 reads; 65,536 bases still refused."""
    cp.check_haplotype_beyond_40k_bases("gpu", TOL, Lh=60_000, R=64)


def test_gpu_window_paired_task_lists_equal_oracle_and_plain_path():
    """k_pair_sort + the PAIRED segments of k_dp forced on small batches (the full-size 100k x 128 test runs them by default): oracle, and the bytes of the plain path."""
    assert len(cp.check_window_pairing("gpu", TOL)) == 5


def test_gpu_lane_per_pair_mapper(monkeypatch):
    """k_kmer_map_lanes forced on small batches (the full-size tests run it by default): positions pair by pair against the oracle's mapper, populate checks."""
    monkeypatch.setenv("OCT_PHMM_LANE_MAPPER", "1")
    cp.check_device_kmer_mapper("gpu", TOL)
    assert cp.check_kmer_mapper_positions("gpu") >= 200
    cp.check_basic("gpu", TOL)
    cp.check_generic_bytes("gpu", TOL)
    cp.check_templates_and_regions("gpu", TOL)
    cp.check_ragged_and_edges("gpu", TOL)
    monkeypatch.setenv("OCT_PHMM_MAP_COUNT_ONLY", "1")
    cp.check_device_kmer_mapper("gpu", TOL)


def test_gpu_canonical_windows_through_lds_and_global_tables_agree():
    assert cp.check_window_tables("gpu", TOL)["n_pairs_shared"] > 20


def test_gpu_upload_refuses_what_the_input_contract_excludes():
    cp.check_input_contract("gpu")


def test_gpu_page_locked_caller_buffers_skip_the_staging_copies():
    assert cp.check_page_locked_caller_buffers("gpu", TOL) == 10


def test_gpu_patched_read_realigner_seam_equals_the_reference_functions():
    """INTEGRATION.md's third seam (read_realigner.cpp:83-155) with its last function replaced by one oct_phmm_align call, linked against liboct_phmm.so
    (prebuilt oracle/_ref/libref_realigner_patched_gpu.so), against the reference's own functions' committed output: region, CIGAR, log-likelihood per read."""
    import check_realigner_patch as cr
    require_reference_build(cr.have("patched_gpu"), "oracle/_ref/libref_realigner_patched_gpu.so")
    assert cr.check("gpu", TOL, golden=True) == 98


def test_gpu_linked_chunks_of_long_reads_and_ragged_reads():
    assert len(cp.check_linked_chunks("gpu", TOL)) == 3


@pytest.mark.parametrize("chunk", ["16", "40"])
def test_gpu_read_records_staged_in_chunks(chunk, monkeypatch):
    monkeypatch.setenv("OCT_PHMM_REC_CHUNK", chunk)
    cp.check_basic("gpu", TOL)
    cp.check_generic_bytes("gpu", TOL)
    cp.check_ragged_and_edges("gpu", TOL)
    cp.check_late_traceback_start("gpu", TOL)
    cp.check_device_kmer_mapper("gpu", TOL)
