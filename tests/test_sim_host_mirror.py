"""CPU suite: the C++ mirror of the reference's HaplotypeLikelihoodArray interface over the C ABI (simulator backend)."""
import check_host_mirror


def test_sim_cpp_host_mirror_matches_oracle_and_maps_errors():
    check_host_mirror.check("sim")
