"""The full-size checks of tests/check_fullsize.py (GPU: tests/test_gpu_fullsize.py) at toy size on the wave simulator, so that the
checking code itself is exercised by the CPU suite."""
import numpy as np
import pytest

import check_fullsize as cf
import oracle
from octopus_amd import synth

pytestmark = pytest.mark.skipif(not oracle.have_ref_array(), reason="oracle/_ref/libref_array.so not built")


def test_sim_fullsize_checks_at_toy_size(monkeypatch):
    monkeypatch.setenv("OCT_PHMM_LATE_MIN_PAIRS", "0")     # the late traceback start and the slice pipeline only engage from 100 k pairs
    monkeypatch.setenv("OCT_PHMM_SLICES", "3")
    rng = np.random.default_rng(3)
    r = cf.check_bench_batch("sim", region=synth.make_region(rng, 30, 4, T=60, Lh=160, B=16, flank=(20, 20), positions="none"))
    assert r["n"] == 120 and r["stats"]["n_dp_traceback"] > 0
    regions = [synth.make_region(rng, R, H, T=50, Lh=150 + 10 * H, B=16, flank=(15, 20), positions="none") for R, H in ((12, 3), (7, 2), (9, 4))]
    r = cf.check_region_stream("sim", regions=regions)
    assert r["regions"] == 3
    monkeypatch.delenv("OCT_PHMM_SLICES")
    g = synth.make_region(rng, 3, 2, T=200, Lh=760, B=256, flank=(60, 60), positions="none", q_values=(8, 15), indels_per_read=2)
    r = cf.check_long_reads("sim", region=g, max_cigar_ops=512)
    assert r["n"] == 6
