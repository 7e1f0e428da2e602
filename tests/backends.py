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
    srcs = list((ROOT / "octopus_amd" / "csrc").glob("*")) + list((ROOT / "tests" / "sim").glob("*.hpp")) + [ROOT / "include" / "oct_phmm.h"]
    if SIM_LIB.exists() and all(SIM_LIB.stat().st_mtime >= s.stat().st_mtime for s in srcs):
        return SIM_LIB
    subprocess.run([CLANG, "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-shared", "-DOCTPHMM_SIM",
                    f"-I{ROOT / 'tests' / 'sim'}", f"-I{ROOT / 'octopus_amd' / 'csrc'}", "-Wno-unused-function",
                    str(ROOT / "octopus_amd" / "csrc" / "oct_phmm.hip"), "-o", str(SIM_LIB)], check=True)
    return SIM_LIB


def make_engine(backend: str, **cfg_kw) -> engine.Engine:
    cfg = abi.Config.default(**cfg_kw)
    if backend == "sim":
        return engine.Engine(cfg, lib_path=build_sim())
    assert backend == "gpu"
    return engine.Engine(cfg)
