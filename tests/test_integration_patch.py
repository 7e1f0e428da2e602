"""CPU suite: the reference's own HaplotypeLikelihoodArray with INTEGRATION.md's patch applied and compiled (oracle/Makefile `patched`),
running on the wave simulator's build of the C ABI, against the unpatched class. See tests/check_integration_patch.py."""
import subprocess

import pytest

import oracle
from backends import ROOT, build_sim


def test_patched_reference_class_equals_the_unpatched_one_through_its_own_accessors():
    if not oracle.have_ref_array():
        pytest.skip("oracle/_ref/libref_array.so not built (no /root/reference here)")
    build_sim()
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "all", "patched"], check=True, stdout=subprocess.DEVNULL)
    assert oracle.have_patched_array("sim")
    import check_integration_patch as ci
    assert ci.check("sim") > 2000


def test_patched_read_assigner_seam_equals_the_reference_functions():
    """The second seam: read_assigner.cpp:145-287 compiled as it is and with its last function replaced by ONE oct_phmm_populate call
    (integration/read_assigner_on_device.inc), on the simulator's build of the C ABI. See tests/check_assigner_patch.py."""
    if not oracle.have_ref_array():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    build_sim()
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "all", "patched"], check=True, stdout=subprocess.DEVNULL)
    import check_assigner_patch as ca
    assert ca.have("ref") and ca.have("patched_sim")
    assert ca.check("sim") > 150


def test_patched_read_realigner_seam_equals_the_reference_functions():
    """The third seam: read_realigner.cpp:83-155 compiled as it is and with its last function replaced by ONE oct_phmm_align call
    (integration/read_realigner_on_device.inc), on the simulator's build of the C ABI: new region, CIGAR and log-likelihood of every read.
    See tests/check_realigner_patch.py."""
    if not oracle.have_ref_array():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    build_sim()
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "all", "patched"], check=True, stdout=subprocess.DEVNULL)
    import check_realigner_patch as cr
    assert cr.have("ref") and cr.have("patched_sim")
    assert cr.check("sim") == 101                    # 98 + three reads on a 65,900-base haplotype, which the device path refuses (the seam's fallback)
    assert cr.check("sim", golden=True) == 98        # (the committed goldens the GPU box compares with are the reference's answers of today)
