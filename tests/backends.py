"""Two ways to execute the product's C ABI in tests:
  * "gpu": octopus_amd/liboct_phmm.so on a real MI355X (tests marked @pytest.mark.gpu)
  * "sim": the SAME host + kernel source compiled for the host against tests/sim (a lockstep wave64 simulator),
           so device-code logic is covered by the CPU suite. Test infrastructure only.
"""
import subprocess
from pathlib import Path

from octopus_amd import abi, engine

ROOT = Path(__file__).resolve().parents[1]
SIM_LIB = ROOT / "tests" / "sim" / "libphmm_sim.so"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build_sim() -> Path:
    srcs = sorted((ROOT / "octopus_amd" / "csrc").glob("*.h*")) + sorted((ROOT / "tests" / "sim").glob("*.hpp")) + [ROOT / "include" / "oct_phmm.h"]
    flags = ["-x", "c++", "-std=c++17", "-O2", "-fPIC", "-shared", "-DOCTPHMM_SIM", "-Wno-unused-function"]
    digest = engine.source_digest(srcs, flags)                 # content, not mtimes: see engine.build
    if engine.up_to_date(SIM_LIB, digest):
        return SIM_LIB
    subprocess.run([CLANG] + flags + [f"-I{ROOT / 'tests' / 'sim'}", f"-I{ROOT / 'octopus_amd' / 'csrc'}",
                                      str(ROOT / "octopus_amd" / "csrc" / "oct_phmm.hip"), "-o", str(SIM_LIB)], check=True)
    engine.stamp(SIM_LIB, digest)
    return SIM_LIB


def make_engine(backend: str, **cfg_kw) -> engine.Engine:
    cfg = abi.Config.default(**cfg_kw)
    if backend == "sim":
        return engine.Engine(cfg, lib_path=build_sim())
    assert backend == "gpu"
    return engine.Engine(cfg)


def require_reference_build(present: bool, what: str) -> None:
    """GPU tests that compare with the REFERENCE's compiled code (oracle/_ref, prebuilt where /root/reference exists and shipped with the
    tree) FAIL when that library is missing: a box fed from a tree whose oracle build broke must not turn its parity tests into skips."""
    assert present, (f"{what} is missing on this box: build it where /root/reference exists "
                     "(python -c 'import __graft_entry__ as g; g.build()') and ship it with the tree")
