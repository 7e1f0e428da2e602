"""CPU suite: oct_phmm_align (best alignment + CIGAR per pair) on the wave simulator vs the oracle."""
import check_align as ca


def test_sim_align_basic():
    ca.check_align_basic("sim")


def test_sim_align_positions_options_lanes_and_bands():
    ca.check_align_positions_and_options("sim")


def test_sim_align_errors():
    ca.check_align_errors("sim")


def test_sim_align_candidate_counts_name_the_saturated_pairs():
    assert ca.check_align_candidate_counts("sim")
