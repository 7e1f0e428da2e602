"""The oracle's evidence chain rebuilds from a clean checkout in one command, and says so loudly when it does not: the working tree's source files
(tracked, plus untracked ones git would not ignore) are copied into a temporary directory together with the two builds of the C ABI the
patched seams link against (the product library and the simulator's - built by engine.build() / tests/backends.build_sim(), which have their own
tests), and oracle.build() must leave every library of oracle/Makefile's `expected` list there. Needs /root/reference (skipped on the GPU box)."""
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

from backends import ROOT, build_sim
from octopus_amd import engine

REF = Path("/root/reference/src/core/models/pairhmm")
ALL_LIBS = ["liboracle.so", "_ref/libref_phmm.so", "_ref/libref_array.so", "_ref/libref_array_avx2.so", "_ref/libref_assigner.so",
            "_ref/libref_array_patched_sim.so", "_ref/libref_assigner_patched_sim.so", "_ref/libref_array_patched_gpu.so",
            "_ref/libref_assigner_patched_gpu.so", "_ref/libref_realigner.so", "_ref/libref_realigner_patched_sim.so", "_ref/libref_realigner_patched_gpu.so"]


def tree_source_files():
    """The tree's source files: what git tracks plus what it would not ignore; in an export without .git (`git archive`, the GPU box's snapshot) the
    directory is walked instead, minus what .gitignore names (built libraries, caches, scratch)."""
    r = subprocess.run(["git", "ls-files", "-co", "--exclude-standard"], cwd=ROOT, capture_output=True, text=True)
    if r.returncode == 0 and r.stdout.strip():
        return r.stdout.split("\n")
    skip_dirs = {".git", "__pycache__", ".pytest_cache", ".hypothesis", "gpurun_out", ".gpurun", "build", ".claude", "_ref"}
    skip_suffix = (".so", ".o", ".a", ".hsaco", ".co", ".pyc", ".srchash")
    skip_files = {"tools/valu_ubench", "tools/urem_probe", "tools/region_calls_bench", "tests/host/host_mirror_sim", "tests/host/host_mirror_gpu", "PROGRESS.jsonl", "COPYCHECK.json"}
    out = []
    for p in ROOT.rglob("*"):
        rel = p.relative_to(ROOT)
        if not p.is_file() or any(part in skip_dirs for part in rel.parts) or p.name.endswith(skip_suffix) or str(rel) in skip_files:
            continue
        out.append(str(rel))
    return out


def run_build(tree: Path) -> subprocess.CompletedProcess:
    return subprocess.run([sys.executable, "-c", "import oracle; oracle.build()"], cwd=tree, capture_output=True, text=True)


@pytest.mark.skipif(not REF.exists(), reason="no /root/reference here")
def test_oracle_builds_from_a_clean_checkout_and_fails_loudly(tmp_path):
    engine.build()
    build_sim()
    files = tree_source_files()
    for f in filter(None, files):
        if (ROOT / f).is_file():
            (tmp_path / f).parent.mkdir(parents=True, exist_ok=True)
            shutil.copy2(ROOT / f, tmp_path / f)
    assert not (tmp_path / "oracle" / "_ref").exists() and not (tmp_path / "oracle" / "liboracle.so").exists()
    for lib in (engine.LIB_PATH, ROOT / "tests" / "sim" / "libphmm_sim.so"):
        shutil.copy2(lib, tmp_path / lib.relative_to(ROOT))
    r = run_build(tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    missing = [name for name in ALL_LIBS if not (tmp_path / "oracle" / name).exists()]
    assert not missing, missing
    # the model header carries every seam's friend line whatever the order the seams were built in
    hpp = (tmp_path / "oracle" / "_ref" / "patched" / "core" / "models" / "haplotype_likelihood_model.hpp").read_text()
    assert hpp.count("friend ") >= 3 and "friend class HaplotypeLikelihoodArray;" in hpp and "friend struct octopus::ReadAssignerDevice;" in hpp
    assert "friend struct octopus::ReadRealignerDevice;" in hpp

    # one statement less in a seam's patch: the build must RAISE (last round it reported success and left no patched libraries)
    inc = tmp_path / "integration" / "read_assigner_on_device.inc"
    lines = inc.read_text().split("\n")
    k = next(i for i, line in enumerate(lines) if line.strip().endswith(";") and "prior_r.insert" in line)
    inc.write_text("\n".join(lines[:k] + ["    this_is_not_declared_anywhere();"] + lines[k + 1:]))
    r = run_build(tmp_path)
    assert r.returncode != 0 and "CalledProcessError" in r.stderr
