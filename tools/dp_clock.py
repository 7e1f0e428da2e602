#!/usr/bin/env python3
"""Shader clock the chip sustains while the DP kernels run: a one-wave probe kernel (tools/clock_probe.hip) samples
s_memtime against the constant reference clock on its own stream while another host thread keeps the 100k x 128 step running (single slice, so
that a window is dominated by one kernel at a time), and while the chip is otherwise idle.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/clock_probe.hip -o tools/libclock_probe.so && python tools/dp_clock.py"""
import ctypes as C
import json
import os
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from octopus_amd import abi, engine, synth   # noqa: E402

probe = C.CDLL(str(ROOT / "tools" / "libclock_probe.so"))
probe.probe_clock.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_double)]


def sample(ms, n):
    out = []
    for _ in range(n):
        g = C.c_double(0)
        assert probe.probe_clock(0, ms, C.byref(g)) == 0
        out.append(g.value)
    return out


res = {"idle_ghz": sample(5.0, 5)}
os.environ["OCT_PHMM_ENV_SWITCHES"] = "1"; os.environ["OCT_PHMM_SLICES"] = "1"
eng = engine.Engine(abi.Config.default(max_indel_error=16))
rb = eng.upload(synth.config_batch("100kx128", seed=42, B=16, positions="none"))
rb.run(); rb.wait()
stop = False


def loop():
    while not stop:
        rb.run(); rb.wait()


t = threading.Thread(target=loop); t.start()
time.sleep(0.3)
res["under_dp_step_ghz_5ms_windows"] = sample(5.0, 40)        # a step is ~32 ms, 72 % of it DP kernels: most 5 ms windows lie inside one
res["under_dp_step_ghz_100ms_window"] = sample(100.0, 3)
stop = True; t.join()
rb.free(); eng.close()
w = sorted(res["under_dp_step_ghz_5ms_windows"])
res["summary"] = {"idle": sum(res["idle_ghz"]) / len(res["idle_ghz"]), "dp_median": w[len(w) // 2], "dp_min": w[0], "dp_max": w[-1],
                  "dp_100ms_mean": sum(res["under_dp_step_ghz_100ms_window"]) / 3}
print(json.dumps(res))
