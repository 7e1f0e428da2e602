"""CPU suite: the real kernel source (octopus_amd/csrc/phmm_kernels.hpp) executed by the lockstep wave simulator,
checked against the reference's golden vectors and the oracle. No GPU needed."""
import pytest

import check_l1


def test_sim_golden_vectors(golden_records):
    assert check_l1.check_golden("sim", golden_records) == 22


@pytest.mark.parametrize("band,n", [(8, 24), (16, 24), (32, 10), (64, 6)])
def test_sim_random_windows_fast_kernel(band, n):
    check_l1.check_random("sim", band, n, seed=100 + band, with_n=False)


@pytest.mark.parametrize("band,n", [(8, 16), (16, 16), (32, 6)])
def test_sim_random_windows_generic_kernel(band, n):
    check_l1.check_random("sim", band, n, seed=200 + band, with_n=True)


def test_sim_unmasked_overload():
    check_l1.check_random("sim", 16, 10, seed=7, masked=False, with_n=False)


def test_sim_int16_overflow_wraps_like_reference():
    check_l1.check_random("sim", 16, 8, seed=9, t_lo=150, t_hi=151, q_max=125, junk=True, with_n=False)


def test_sim_int32_lanes_golden_vectors(golden_records):
    assert check_l1.check_golden("sim", golden_records, score_bits=32) == 17


@pytest.mark.parametrize("band,n", [(8, 10), (16, 10), (32, 4), (64, 3)])
def test_sim_int32_lanes_random_windows(band, n):
    check_l1.check_random("sim", band, n, seed=300 + band, with_n=True, score_bits=32)


def test_sim_int32_lanes_do_not_wrap_where_int16_does():
    check_l1.check_random("sim", 16, 4, seed=9, t_lo=150, t_hi=151, q_max=125, junk=True, with_n=False, score_bits=32)


def test_sim_exact_add_mode_gives_same_results(monkeypatch):
    """The host picks v_add_u32 adds only when its bounds prove no lane can wrap; forcing the v_pk_add_u16 path must not change anything."""
    monkeypatch.setenv("OCT_PHMM_EXACT_ADDS", "1")
    check_l1.check_random("sim", 16, 10, seed=116, with_n=False)
    check_l1.check_random("sim", 8, 8, seed=208, with_n=True)


@pytest.mark.parametrize("band,bits,n", [(128, 16, 4), (128, 32, 3), (256, 32, 2), (256, 16, 2)])
def test_sim_wide_bands_streaming_kernel(band, bits, n):
    check_l1.check_random("sim", band, n, seed=400 + band + bits, t_lo=40, t_hi=200, with_n=True, score_bits=bits)


@pytest.mark.parametrize("band,n", [(128, 4), (256, 3)])
def test_sim_multi_wave_kernel_fast_cost_and_one_wave_form(band, n, monkeypatch):
    """Bands 128 / 256 on int32 lanes run one task per workgroup (k_dp_mw: B / 64 waves, border cells through LDS mailboxes); pure-ACGT windows
    take its fast-cost form. OCT_PHMM_MULTI_WAVE=0 keeps the one-wave-per-task kernel (k_dp_wide) covered."""
    check_l1.check_random("sim", band, n, seed=500 + band, t_lo=30, t_hi=300, with_n=False, score_bits=32)
    check_l1.check_random("sim", band, 2, seed=520 + band, t_lo=200, t_hi=330, q_max=125, junk=True, with_n=False, score_bits=32)   # unrelated sequences: walks that wander over the band
    monkeypatch.setenv("OCT_PHMM_MW_PLANES", "1")         # all planes of a task in one wave (what a launch of >= 640 tasks takes)
    check_l1.check_random("sim", band, n, seed=530 + band, t_lo=30, t_hi=300, with_n=False, score_bits=32)
    check_l1.check_random("sim", band, 2, seed=540 + band, t_lo=40, t_hi=200, with_n=True, score_bits=32)
    monkeypatch.delenv("OCT_PHMM_MW_PLANES")
    monkeypatch.setenv("OCT_PHMM_MULTI_WAVE", "0")
    monkeypatch.setenv("OCT_PHMM_WALK_STAGE", "0")        # ... and the lockstep walker instead of k_walk_long
    check_l1.check_random("sim", band, 2, seed=600 + band, t_lo=40, t_hi=200, with_n=True, score_bits=32)
