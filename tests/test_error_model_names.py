"""The reference's NAMED error models (--sequence-error-model, error_model_factory.cpp:220-517): every (library preparation, sequencer) pair the factory
holds resolves through oct_phmm_error_model_by_name / _by_label to the factory's tables, and the six penalty vectors the product makes from them equal the
reference's own BasicRepeatBasedIndelErrorModel / BasicRepeatBasedSNVErrorModel (compiled in place) given the same tables. The tables come from
tests/golden/error_model_tables.json (tools/make_error_model_tables.py); where /root/reference exists the fixture is re-derived from the source first."""
import ctypes as C
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle
from backends import build_sim
from octopus_amd import abi, engine

ROOT = Path(__file__).resolve().parents[1]
FIX = ROOT / "tests" / "golden" / "error_model_tables.json"
LIBS = ["pcr", "pcr_free", "tenx", "mda"]
SEQS = ["hiseq_2000", "hiseq_2500", "hiseq_4000", "xten", "novaseq", "bgiseq_500", "pacbio", "pacbio_ccs"]


@pytest.fixture(scope="module")
def tables(tmp_path_factory):
    if Path("/root/reference/src/core/models/error/error_model_factory.cpp").exists():
        # re-derived into a scratch directory and compared: the source tree is never written by a test run (read-only checkouts, pytest-xdist)
        out = tmp_path_factory.mktemp("error_model_tables")
        subprocess.run([sys.executable, str(ROOT / "tools" / "make_error_model_tables.py"), f"--out-dir={out}"], check=True, capture_output=True)
        hdr = Path("octopus_amd") / "csrc" / "phmm_error_model_tables.hpp"
        assert (out / "tests" / "golden" / "error_model_tables.json").read_text() == FIX.read_text() and (out / hdr).read_text() == (ROOT / hdr).read_text(), \
            "committed tables are not what the reference's factory holds (python tools/make_error_model_tables.py regenerates them)"
    return json.loads(FIX.read_text())


def expand(row, n):
    return [row[i] if i < len(row) else row[-1] for i in range(n)]


def test_every_factory_name_resolves_to_the_factory_tables(tables):
    lib_path = build_sim()
    n = 0
    for lib in LIBS:
        for seq in SEQS:
            key = f"{lib}.{seq}"
            for lname in tables["library_names"][lib]:
                for sname in tables["sequencer_names"][seq]:
                    if key not in tables["indel_open"]:
                        for spell in (lname, lname.lower()):
                            with pytest.raises(engine.EngineError):
                                engine.error_model_by_name(spell, sname, lib_path)
                        continue
                    rows = tables["indel_open"][key]
                    for m in (engine.error_model_by_name(lname, sname, lib_path), engine.error_model_by_name(lname.lower(), sname.capitalize(), lib_path),
                              engine.error_model_by_label(f"{lname}.{sname}", lib_path)):
                        got = [list(m.at_homopolymer_open), list(m.cg_homopolymer_open), list(m.dinucleotide_open), list(m.trinucleotide_open)]
                        assert got == [expand(r, 50) for r in rows], key
                        assert [list(m.homopolymer_extend), list(m.dinucleotide_extend), list(m.trinucleotide_extend)] == [expand(r, 50) for r in tables["extend"]]
                        assert [list(m.snv_caps[k]) for k in range(3)] == [expand(r, 51) for r in tables["snv_caps"][lib]]
                        assert m.use_snv_model == (0 if seq in tables["no_snv_model_sequencers"] else 1)
                        n += 1
    assert n == 3 * 28 + 3 * 8                                    # PCR-free has two spellings
    # defaults for missing parts (parse_model_config): "" / None -> PCR-free, HiSeq-2500
    d = engine.default_error_model(lib_path)
    for m in (engine.error_model_by_name(None, None, lib_path), engine.error_model_by_label("", lib_path), engine.error_model_by_label("PCR-free", lib_path),
              engine.error_model_by_label(".HiSeq-2500", lib_path)):
        assert bytes(m) == bytes(d)
    assert tables["default"] == "pcr_free.hiseq_2500"
    for bad in (("TruSeq", None), (None, "MiSeq"), ("PCR.", "X10")):
        with pytest.raises(engine.EngineError):
            engine.error_model_by_name(bad[0], bad[1], lib_path)
    with pytest.raises(engine.EngineError):
        engine.error_model_by_label("10X.PacBio", lib_path)


@pytest.mark.skipif(not oracle.have_ref(), reason="reference build absent")
def test_vectors_of_every_named_model_equal_the_reference_classes(tables):
    """For each distinct parameter set: the product's six vectors (oct_phmm_penalty_vectors) against the reference's model classes built from the same
    (unexpanded) factory rows, on strings with planted repeats of periods 1-6, Ns and substitution masks. PacBio sets: indel vectors only (no SNV model)."""
    from check_error_model import corpus
    lib_path = build_sim()
    seqs, subs, bases, off = corpus(seed=5, n_strings=120)
    seen = {}
    for lib in LIBS:
        for seq in SEQS:
            key = f"{lib}.{seq}"
            if key not in tables["indel_open"]:
                continue
            sig = json.dumps([tables["indel_open"][key], tables["snv_caps"][lib], seq in tables["no_snv_model_sequencers"]])
            if sig in seen:
                continue
            seen[sig] = key
            m = engine.error_model_by_name(tables["library_names"][lib][0], tables["sequencer_names"][seq][0], lib_path)
            got = engine.penalty_vectors(m, bases, off, np.concatenate(subs), lib_path=lib_path)
            tabs = tables["indel_open"][key] + tables["extend"] + tables["snv_caps"][lib]
            flat = np.asarray([v for t in tabs for v in t], np.int8); lens = np.asarray([len(t) for t in tabs], np.uint32)
            p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
            for i, (s, sub) in enumerate(zip(seqs, subs)):
                n = len(s)
                want = [np.zeros(n, np.int8), np.zeros(n, np.int8), np.zeros(n, np.uint8), np.zeros(n, np.int8), np.zeros(n, np.uint8), np.zeros(n, np.int8)]
                oracle.ref().ref_error_models(p(flat), p(lens), bytes(s), n, p(np.ascontiguousarray(sub, dtype=np.uint8)), *[p(w) for w in want])
                mine = [g[off[i]:off[i + 1]] for g in got]
                which = range(6) if m.use_snv_model else range(2)
                for k in which:
                    assert np.array_equal(np.asarray(mine[k]).view(np.uint8), want[k].view(np.uint8)), (key, i, k)
                if not m.use_snv_model:                          # model.cpp:69-73: masks = the haplotype itself, priors = 100
                    assert bytes(np.asarray(mine[2])) == bytes(s) and bytes(np.asarray(mine[4])) == bytes(s)
                    assert set(np.asarray(mine[3]).tolist()) <= {100} and set(np.asarray(mine[5]).tolist()) <= {100}
    assert len(seen) >= 10
