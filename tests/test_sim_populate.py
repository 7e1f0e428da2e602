"""CPU suite: the whole populate() pipeline (host API + every kernel) on the wave simulator vs the oracle, bit for bit."""
import pytest
import check_populate as cp


def test_sim_populate_basic():
    cp.check_basic("sim")


def test_sim_populate_generic_bytes():
    s = cp.check_generic_bytes("sim")
    assert s["n_dp_score_only"] + s["n_dp_traceback"] > 0


def test_sim_populate_templates_and_regions():
    cp.check_templates_and_regions("sim")


def test_sim_populate_ragged_reads_edges_and_short_haplotype():
    cp.check_ragged_and_edges("sim")
    cp.check_one_shot_populate_streams_slices_back("sim")


def test_sim_populate_mapping_quality_and_flank_options():
    cp.check_mapping_quality_options("sim")


def test_sim_device_kmer_mapper_matches_reference_mapper():
    cp.check_device_kmer_mapper("sim")


def test_sim_device_mapper_positions_equal_the_oracle_mapper_on_tie_rich_haplotypes(monkeypatch):
    assert cp.check_kmer_mapper_positions("sim") >= 200
    monkeypatch.setenv("OCT_PHMM_BIG_MAPPER", "1")
    assert cp.check_kmer_mapper_positions("sim", seeds=(71, 72)) >= 50    # the one-workgroup-per-pair mapper of very long haplotypes


def test_sim_populate_int32_lanes():
    a, b = cp.check_int32_lanes("sim")
    assert a["n_dp_traceback"] > 0 and b["n_dp_score_only"] + b["n_dp_traceback"] > 0


def test_sim_populate_wide_bands_and_long_reads():
    a, b = cp.check_wide_and_long("sim")
    assert a["n_dp_score_only"] + a["n_dp_traceback"] > 0 and b["n_dp_score_only"] + b["n_dp_traceback"] > 0


def test_sim_big_haplotype_mapper_matches_reference_mapper(monkeypatch):
    monkeypatch.setenv("OCT_PHMM_BIG_MAPPER", "1")      # force the one-workgroup-per-pair mapper used for very long haplotypes
    cp.check_device_kmer_mapper("sim")


def test_sim_haplotypes_of_40k_bases_and_more_map_on_the_device():
    cp.check_haplotype_beyond_40k_bases("sim")


def test_sim_populate_in_slices(monkeypatch):
    """Large batches are cut into slices of whole haplotypes that run on separate streams; force 3 slices on small batches."""
    monkeypatch.setenv("OCT_PHMM_SLICES", "3")
    cp.check_basic("sim")
    cp.check_templates_and_regions("sim")
    cp.check_device_kmer_mapper("sim")
    cp.check_ragged_and_edges("sim")


def test_sim_long_reads_at_narrow_bands_stream():
    stats = cp.check_long_reads_at_narrow_bands("sim", T=700, Lh=1900, n_reads=3)
    assert all(s["n_dp_score_only"] + s["n_dp_traceback"] > 0 for s in stats) and any(s["n_dp_traceback"] > 0 for s in stats)


def test_sim_empty_batches():
    cp.check_empty_batches("sim")


def test_sim_late_traceback_start_equals_oracle_and_plain_path():
    cp.check_late_traceback_start("sim")


def test_sim_random_scenarios_with_late_traceback_start(monkeypatch):
    import check_fuzz
    monkeypatch.setenv("OCT_PHMM_LATE_MIN_PAIRS", "0")
    assert check_fuzz.check_fuzz("sim", seed=99, n=8) == 8


def test_sim_chunked_traceback_launches():
    cp.check_chunked_traceback("sim")


def test_sim_matrix_equals_the_reference_array_populate():
    import oracle
    import pytest
    if not oracle.have_ref_array():
        pytest.skip("oracle/_ref/libref_array.so not built (no /root/reference)")
    assert cp.check_against_reference_array("sim") > 400


def test_sim_streamed_upload_and_growing_calls():
    cp.check_streamed_upload_and_growing_calls("sim")



def test_sim_leading_slice_without_pairs_still_hashes_the_reads():
    cp.check_leading_haplotypes_without_reads("sim")


def test_sim_pairs_with_equal_candidates_share_one_result():
    assert len(cp.check_shared_pairs("sim")) == 3       # (one slice with given and with device positions, three slices with device positions)


def test_sim_reads_beyond_32k_bases():
    """Round 6: reads of 32,768 bases and more through all three long-read walkers (20-bit window coordinates in the queued walk events); 2^20 is still refused."""
    assert len(cp.check_reads_beyond_32k_bases("sim")) == 3
    import check_align as ca
    ca.check_align_reads_beyond_32k_bases("sim")


def test_sim_device_sized_and_host_sized_launches_agree():
    assert cp.check_launch_modes("sim") == 6


def test_sim_populate_host_sized_launches(monkeypatch):
    """Region-sized batches run with device-sized launches by default; the single-slice host-sized path (the mid-step read-back) on the same checks."""
    monkeypatch.setenv("OCT_PHMM_DEVICE_SIZED", "0")
    cp.check_basic("sim")
    cp.check_generic_bytes("sim")
    cp.check_templates_and_regions("sim")
    cp.check_ragged_and_edges("sim")


@pytest.mark.parametrize("mode", ["0", "1"])
def test_sim_populate_lockstep_walkers(mode, monkeypatch):
    """Small traceback launches give every walk a 16-lane row (k_walk_rows); the lockstep walker - one line per lane in registers (0: what big launches use) or
    tiles staged in LDS (1) - on the same checks."""
    monkeypatch.setenv("OCT_PHMM_WALK_STAGE", mode)
    cp.check_basic("sim")
    cp.check_templates_and_regions("sim")
    if mode == "0":
        cp.check_late_traceback_start("sim")        # (early-stopping walks are the register walker's: what launches of > 64 k walks take)


def test_sim_scratch_allocation_failures_fall_back_and_leave_no_error():
    cp.check_scratch_allocation_failures("sim")


def test_sim_mapper_mismatch_account_feeds_the_fast_path():
    stats = cp.check_mapper_mismatch_account("sim")
    assert len(stats) == 5


def test_sim_lane_per_pair_mapper(monkeypatch):
    """Big batches map with one lane per (haplotype, read) pair (k_kmer_map_lanes: the exact shortcut per lane, the undecided pairs counted by the whole wave);
    forced on the small batches of the mapper and populate checks."""
    monkeypatch.setenv("OCT_PHMM_LANE_MAPPER", "1")
    cp.check_device_kmer_mapper("sim")
    assert cp.check_kmer_mapper_positions("sim") >= 200
    cp.check_basic("sim")
    cp.check_generic_bytes("sim")
    cp.check_templates_and_regions("sim")
    cp.check_ragged_and_edges("sim")
    monkeypatch.setenv("OCT_PHMM_SLICES", "3")
    cp.check_device_kmer_mapper("sim")
    monkeypatch.setenv("OCT_PHMM_MAP_COUNT_ONLY", "1")              # every pair through the lane kernel's counting path
    cp.check_device_kmer_mapper("sim")


def test_sim_window_paired_task_lists_equal_oracle_and_plain_path():
    assert len(cp.check_window_pairing("sim")) == 5


def test_sim_page_locked_caller_buffers_skip_the_staging_copies():
    assert cp.check_page_locked_caller_buffers("sim") == 10


def test_sim_linked_chunks_of_long_reads_and_ragged_reads():
    stats = cp.check_linked_chunks("sim")
    assert len(stats) == 3 and all(s["n_dp_score_only"] + s["n_dp_traceback"] > 0 for s in stats)


@pytest.mark.parametrize("chunk", ["16", "40"])
def test_sim_read_records_staged_in_chunks(chunk, monkeypatch):
    """The packed int16 DP kernels keep a chunk of iterations' read records in LDS and restage at its borders where whole reads would leave one wave per SIMD
    (500-base reads); forced on the small reads of the populate checks (borders inside the rolling initialisation, the capture phase, the late traceback start)."""
    monkeypatch.setenv("OCT_PHMM_REC_CHUNK", chunk)
    cp.check_basic("sim")
    cp.check_generic_bytes("sim")
    cp.check_ragged_and_edges("sim")
    if chunk == "16":
        cp.check_late_traceback_start("sim")
        cp.check_templates_and_regions("sim")


def test_sim_upload_refuses_what_the_input_contract_excludes():
    cp.check_input_contract("sim")


def test_sim_canonical_windows_through_lds_and_global_tables_agree():
    assert cp.check_window_tables("sim")["n_pairs_shared"] > 20


def test_sim_a_full_device_makes_a_pool_trim_its_siblings_and_the_call_still_answers():
    """ADVICE r04 / round 5's fuzz: a handle whose device allocation fails gives back its own cached blocks, then every sibling handle's (DevPool::trim_device), and retries - instead of
    reporting out-of-memory while other handles sit on free blocks. The simulator's allocator is told to fail the next two big allocations (the first attempt and the retry behind the pool's
    own trim; the one behind the siblings' trim succeeds) while two handles hold cached blocks; the call must still equal the oracle; with more failures sprinkled over a call it either still answers right or reports the HIP error, and the handle stays usable."""
    import ctypes as C
    import numpy as np
    import oracle
    from backends import build_sim, make_engine
    from octopus_amd import abi, synth
    lib = C.CDLL(str(build_sim()))
    lib.octsim_fail_next_mallocs.argtypes = [C.c_size_t, C.c_long]; lib.octsim_fail_next_mallocs.restype = C.c_long
    rng = np.random.default_rng(3)
    b1 = synth.batch_from_regions([synth.make_region(rng, 30, 4, T=60, Lh=180, B=8, positions="none")])
    b2 = synth.batch_from_regions([synth.make_region(rng, 90, 6, T=70, Lh=200, B=8, positions="none")])
    e1, e2 = make_engine("sim", max_indel_error=8), make_engine("sim", max_indel_error=8)
    for _ in range(2):
        e1.populate(b1); e2.populate(b2)                # both pools hold cached blocks now
    want, _, _ = oracle.populate(abi.Config.default(max_indel_error=8), b2)
    before = lib.octsim_fail_next_mallocs(1 << 16, 2)
    got, st = e1.populate(b2)                            # a batch of another size on e1: a new block is needed
    assert st.code == abi.OK and np.array_equal(got, want)
    assert lib.octsim_fail_next_mallocs(0, 0) == before + 2       # both failures were met, the third attempt (behind the siblings' trim) got the block
    got2, _ = e2.populate(b2)                            # the sibling whose cache was trimmed carries on
    assert np.array_equal(got2, want)
    b3 = synth.batch_from_regions([synth.make_region(rng, 200, 7, T=80, Lh=230, B=8, positions="none")])
    want3, _, _ = oracle.populate(abi.Config.default(max_indel_error=8), b3)
    lib.octsim_fail_next_mallocs(1 << 12, 5)             # five failures wherever they fall (batch block, task lists, traceback scratch: the scratch falls back to smaller chunks)
    got3, st3 = e1.populate(b3, raise_on_error=False)
    lib.octsim_fail_next_mallocs(0, 0)
    assert st3.code in (abi.OK, abi.EHIP)                # either it found room after trimming, or it says so - never a wrong matrix
    if st3.code == abi.OK:
        assert np.array_equal(got3, want3)
    got4, st4 = e1.populate(b3)                          # ... and the handle is usable afterwards
    assert st4.code == abi.OK and np.array_equal(got4, want3)
    e1.close(); e2.close()
