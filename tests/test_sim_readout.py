"""CPU suite: the genotype read-out kernels on the wave simulator vs the oracle's ConstantMixtureGenotypeLikelihoodModel."""
import check_readout as cr


def test_sim_genotype_readout_all_ploidies_and_zygosities():
    cr.check_readout("sim")


def test_sim_genotype_readout_rejects_bad_input():
    cr.check_readout_errors("sim")
