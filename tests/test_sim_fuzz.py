"""CPU suite: randomised end-to-end scenarios on the wave simulator vs the oracle."""
import check_fuzz


def test_sim_random_scenarios():
    assert check_fuzz.check_fuzz("sim", seed=2024, n=14) == 14
