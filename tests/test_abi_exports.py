"""The product C-ABI library builds for gfx950 here (hipcc cross-compiles without a GPU), loads, and exports every
function include/oct_phmm.h declares. No compute call is made: without a device only the device-free entry points run."""
import ctypes as C
import re
from pathlib import Path

import pytest

from octopus_amd import abi, engine

ROOT = Path(__file__).resolve().parents[1]


def declared_functions():
    text = (ROOT / "include" / "oct_phmm.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(oct_phmm_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    engine.build()
    return C.CDLL(str(ROOT / "octopus_amd" / "liboct_phmm.so"))


def test_header_declares_the_boundary():
    names = declared_functions()
    for need in ("oct_phmm_create", "oct_phmm_destroy", "oct_phmm_populate", "oct_phmm_batch_upload", "oct_phmm_batch_run",
                 "oct_phmm_batch_wait", "oct_phmm_batch_download", "oct_phmm_align_windows", "oct_phmm_align",
                 "oct_phmm_batch_genotype_likelihoods", "oct_phmm_server_create", "oct_phmm_server_populate"):
        assert need in names


def test_library_exports_every_declared_symbol(lib):
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_device_free_entry_points(lib):
    cfg = abi.Config.default()
    assert cfg.struct_size == C.sizeof(abi.Config)
    c2 = abi.Config()
    lib.oct_phmm_config_default(C.byref(c2))
    assert bytes(c2) == bytes(abi.Config.default())
    lib.oct_phmm_strerror.restype = C.c_char_p
    assert lib.oct_phmm_strerror(0) == b"ok"
    assert b"too large" in lib.oct_phmm_strerror(abi.EBAND)
    assert b"too short" in lib.oct_phmm_strerror(abi.ESHORT_HAPLOTYPE)


def test_create_fails_loudly_without_a_device(lib):
    if lib.oct_phmm_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    cfg = abi.Config.default()
    rc = lib.oct_phmm_create(C.byref(cfg), C.byref(h))
    assert rc in (abi.ENODEVICE, abi.EHIP) and not h.value
    with pytest.raises(Exception):
        engine.Engine(cfg)


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/oct_phmm.h compiles as C99 (no C++ in the signatures) and links against the library."""
    import subprocess
    engine.build()
    src = tmp_path / "cabi.c"
    src.write_text('#include "oct_phmm.h"\n'
                   'int main(void) { oct_phmm_config c; oct_phmm_config_default(&c); return c.struct_size == sizeof c && oct_phmm_device_count() >= 0 ? 0 : 1; }\n')
    exe = tmp_path / "cabi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{ROOT / 'include'}", str(src),
                    f"-L{ROOT / 'octopus_amd'}", "-loct_phmm", f"-Wl,-rpath,{ROOT / 'octopus_amd'}", "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0
