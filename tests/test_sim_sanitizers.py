"""The real host + kernel source, compiled for the host with AddressSanitizer and UBSan, through populate / align / read-out on the
wave simulator ("device" memory is plain malloc there, so any out-of-bounds access of a kernel or of the host API trips a redzone) -
and with ThreadSanitizer through the region server (callers, two gathering workers, their finisher threads, the per-call wake-ups)."""
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
ce.check_custom_model_file(backends.build_sim(), n_models=30, n_strings=8)     # round 5: the model-file reader (malformed texts too) and its string-keyed look-ups
ce.check_custom_model_in_calls("sim")
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
os.environ["OCT_PHMM_DEVICE_SIZED"] = "0"            # round 6: host-sized launches size the traceback scratch exactly - a row of k_walk_rows' last wave that has no task must not fetch from "its" group
check_fuzz.check_fuzz("sim", seed=77, n=44)        # (scenario 42 of this seed faulted on the GPU: the next-window prefetch of a task-less row, band 64)
os.environ.pop("OCT_PHMM_DEVICE_SIZED", None)
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


TSAN_DRIVER = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from pathlib import Path
import backends
backends.build_sim = lambda: Path({lib!r})
import check_server
calls, batches = check_server.check_server("sim", n_threads=6, per_thread=8)                       # OCT_PHMM_SERVER_WORKERS=2: two workers gather from one queue
assert batches < calls
check_server.check_server("sim", n_threads=4, per_thread=5, seed=23, devices=[0, 0])               # ... and two "devices" with two workers each
check_server.check_server_rejects_malformed_calls("sim")
check_server.check_server_contract_violation_reaches_only_its_caller("sim")
import check_error_model as ce
ce.check_align_and_server_generate_the_vectors("sim")            # error models installed while callers are being served (the workers pick them up between batches)
ce.check_custom_model_in_calls("sim")                            # ... and a model read from a file, shared by reference between the server's handles
print("TSAN-OK")
"""


@pytest.mark.skipif(os.environ.get("OCT_RUN_SANITIZERS") != "1", reason="three minutes (ThreadSanitizer build of the whole library): set OCT_RUN_SANITIZERS=1")
def test_region_server_is_clean_under_thread_sanitizer(tmp_path):
    """Round 5 rebuilt the server around threads (per worker a gatherer and a finisher, callers that compute their regions' input facts and sleep on their own
    condition variable): the simulator's lane coroutines are announced to the tool as fibers (tests/sim/hipsim.hpp), device work takes turns at the server's sim_mu,
    everything else - queues, slots, counters, wake-ups - runs as in the product. Any report fails the test."""
    rt = subprocess.run([CLANG, "-print-file-name=libclang_rt.tsan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not rt or not Path(rt).exists():
        pytest.skip("no ThreadSanitizer runtime next to the ROCm clang")
    lib = tmp_path / "libphmm_sim_tsan.so"
    subprocess.run([CLANG, "-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-DOCTPHMM_SIM", "-fsanitize=thread", "-fno-omit-frame-pointer",
                    f"-I{ROOT / 'tests' / 'sim'}", f"-I{ROOT / 'octopus_amd' / 'csrc'}", "-Wno-unused-function", str(ROOT / "octopus_amd" / "csrc" / "oct_phmm.hip"), "-o", str(lib)], check=True)
    env = dict(os.environ, LD_PRELOAD=rt, TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:history_size=4:second_deadlock_stack=1:exitcode=0",
               OCT_PHMM_ENV_SWITCHES="1", OCT_PHMM_SERVER_WORKERS="2")
    p = subprocess.run([sys.executable, "-c", TSAN_DRIVER.format(root=str(ROOT), tests=str(ROOT / "tests"), lib=str(lib))], env=env, capture_output=True, text=True, timeout=2400)
    assert p.returncode == 0 and "TSAN-OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
    assert "WARNING: ThreadSanitizer" not in p.stderr, p.stderr[:6000]
