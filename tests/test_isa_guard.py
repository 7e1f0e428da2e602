"""CPU suite: the device ISA of the product library holds no 24-bit float-reciprocal integer division.

Round 4's one device fault was the compiler's expansion of `(x >> 8) % n` (operands provably below 2^24: one v_rcp_iflag_f32, a truncated float quotient, one
float compare as the only correction) - it returns 0xffffff for 2.8 % of the dividends at n = 11 on gfx950 (tools/urem_probe.hip,
profiles/r04_step8_urem24_probe.txt), k_window_region lost keys and the region server faulted. Nothing in the C++ shows which expansion a `%` gets, so this
test reads the assembly (hipcc -S --cuda-device-only, ~90 s, cached by source digest): the detector must fire on the probe's kernel - the very expression
that was in the product - and must stay silent on the product. It also prints the kernels that spill registers (the -s output of the suite shows them)."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
import isa_report  # noqa: E402

HIPCC = Path(isa_report.HIPCC)


@pytest.mark.skipif(not HIPCC.exists(), reason="no hipcc here")
def test_detector_fires_on_the_expression_that_broke_round_4():
    kinds = [k for _, _, k in isa_report.divisions(isa_report.device_asm(ROOT / "tools" / "urem_probe.hip"))]
    assert "u24" in kinds, kinds          # `(tag * 0x9e3779b1u >> 8) % n24`: if this stops being flagged the guard below guards nothing


@pytest.mark.skipif(not HIPCC.exists(), reason="no hipcc here")
def test_product_library_has_no_24_bit_division_and_reports_its_spills(capsys):
    asm = isa_report.device_asm(ROOT / "octopus_amd" / "csrc" / "oct_phmm.hip")
    div = isa_report.divisions(asm)
    bad = [(name, i) for name, i, kind in div if kind == "u24"]
    assert not bad, f"24-bit float-reciprocal division expansions (use a multiply-high or widen an operand): {bad}"
    sp = isa_report.spills(asm)
    assert len(sp) >= 150, len(sp)           # the metadata was parsed (175 kernels in round 4)
    spilled = {k: v for k, v in sp.items() if v[0]}
    with capsys.disabled():
        print(f"\n[isa] {len(sp)} kernels, {len(div)} reciprocal divisions (none 24-bit); VGPR spills in {len(spilled)} kernels:")
        for k, (v, s, scratch) in sorted(spilled.items()):
            print(f"[isa]   {v} VGPR, {scratch} B scratch: {k[:120]}")
    # the headline kernels (band 16, packed int16 lanes) must not spill at all: a spill inside their loops is a silent 2x
    for k, (v, s, scratch) in sp.items():
        if "k_dpILi16E" in k or "k_dp_pairILi16E" in k or "k_kmer_map_lanes" in k or "k_classify" in k:
            assert v == 0 and scratch == 0, (k, v, scratch)
    # ... and scratch memory stays where it is known to be (round 5: a six-entry array indexed by a list id put scratch loads and stores into the prologue of EVERY
    # DP and walk kernel - found by this test): the kernels listed below, nobody else
    allowed = ("k_dp_rowsILb1E", "k_dp_wideILi256ELb0ELb1E",                        # the long-read kernels with VGPR spills
               "k_penalty_vectors", "k_genotype_lik", "k_walk_strings")              # per-thread workspaces by design (error model run lists, read-out, the test seam's string walker)
    with_scratch = sorted(k for k, (v, s, scratch) in sp.items() if scratch and not any(a in k for a in allowed))
    assert not with_scratch, with_scratch
    # ... and where a long-read DP kernel does spill, the spill sits in the loop over task groups (once per group of 10^4 iterations), never inside a DP loop
    sites = isa_report.spill_sites(asm)
    for k, (n_scratch, smallest_loop, n_instr) in sites.items():
        if "k_dp" in k:
            assert smallest_loop == 0 or smallest_loop > n_instr // 2, (k, n_scratch, smallest_loop, n_instr)
