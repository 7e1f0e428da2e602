"""The real host + kernel source, compiled for the host with AddressSanitizer and UBSan, through populate / align / read-out on the
wave simulator ("device" memory is plain malloc there, so any out-of-bounds access of a kernel or of the host API trips a redzone)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

DRIVER = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from pathlib import Path
import backends
backends.build_sim = lambda: Path({lib!r})
import check_populate as cp, check_align as ca, check_readout as cr
cp.check_basic("sim"); cp.check_templates_and_regions("sim"); cp.check_ragged_and_edges("sim"); cp.check_device_kmer_mapper("sim")
cp.check_int32_lanes("sim"); cp.check_wide_and_long("sim"); cp.check_long_reads_at_narrow_bands("sim", T=260, Lh=800, n_reads=2)
ca.check_align_basic("sim"); ca.check_align_errors("sim"); cr.check_readout("sim"); cr.check_readout_errors("sim")
import check_fuzz
check_fuzz.check_fuzz("sim", seed=5, n=6)
import check_server
check_server.check_server("sim", n_threads=3, per_thread=5)
cp.check_empty_batches("sim")
cp.check_shared_pairs("sim")                       # canonical windows, matcher, verifier, tables across slices
cp.check_launch_modes("sim")                       # device- and host-sized launches, forced scratch overflow, the multi-region forms (two DP launches, tiled scan, DMA copies)
import check_error_model as ce
ce.check_device_kernels_on_the_corpus("sim", 40)   # both penalty-vector kernels
import os
os.environ["OCT_PHMM_LANE_MAPPER"] = "1"           # round 4: one lane per pair (probes, the pass over the hash rows, the wave's counting of undecided pairs), the mismatch account
cp.check_device_kmer_mapper("sim"); cp.check_ragged_and_edges("sim"); cp.check_mapper_mismatch_account("sim")
os.environ.pop("OCT_PHMM_LANE_MAPPER", None)
os.environ["OCT_PHMM_REC_CHUNK"] = "16"            # read records restaged every 16 iterations (the row pointer is rebased below the row's start)
cp.check_basic("sim"); cp.check_generic_bytes("sim"); cp.check_late_traceback_start("sim")
os.environ.pop("OCT_PHMM_REC_CHUNK", None)
cp.check_linked_chunks("sim")                      # templates of linked chunks, ragged reads
cp.check_page_locked_caller_buffers("sim")         # page-locked arrays / out: the direct upload path, results landing in the caller's buffer
cp.check_window_tables("sim")                      # canonical windows: the table of a region in LDS (one and two key classes) and the tables in global memory
cp.check_input_contract("sim")                     # what an upload refuses, single- and multi-threaded checks
print("SANITIZED-OK")
"""


@pytest.mark.skipif(os.environ.get("OCT_RUN_SANITIZERS") != "1", reason="two minutes (sanitizer build of the whole library): set OCT_RUN_SANITIZERS=1")
def test_sim_build_is_clean_under_asan_and_ubsan(tmp_path):
    rt = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not rt or not Path(rt).exists():
        pytest.skip("no ASan runtime next to the ROCm clang")
    lib = tmp_path / "libphmm_sim_asan.so"
    subprocess.run([CLANG, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-DOCTPHMM_SIM", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", f"-I{ROOT / 'tests' / 'sim'}", f"-I{ROOT / 'octopus_amd' / 'csrc'}",
                    "-Wno-unused-function", str(ROOT / "octopus_amd" / "csrc" / "oct_phmm.hip"), "-o", str(lib)], check=True)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([sys.executable, "-c", DRIVER.format(root=str(ROOT), tests=str(ROOT / "tests"), lib=str(lib))],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0 and "SANITIZED-OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
