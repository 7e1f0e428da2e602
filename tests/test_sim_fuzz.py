"""CPU suite: randomised end-to-end scenarios on the wave simulator vs the oracle."""
import check_fuzz


def test_sim_random_scenarios():
    assert check_fuzz.check_fuzz("sim", seed=2024, n=14) == 14


def test_sim_random_scenarios_with_pair_sharing_forced_on(monkeypatch):
    """The same generator with exact de-duplication of pairs switched on for every (small) batch and three slices, so that tables travel
    between slices; a second pass with both hashes cut to three bits (collisions everywhere)."""
    monkeypatch.setenv("OCT_PHMM_DEDUP", "1")
    monkeypatch.setenv("OCT_PHMM_SLICES", "3")
    assert check_fuzz.check_fuzz("sim", seed=9100, n=6) == 6
    monkeypatch.setenv("OCT_PHMM_DEDUP_HASH_BITS", "3")
    assert check_fuzz.check_fuzz("sim", seed=9101, n=6) == 6
