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
