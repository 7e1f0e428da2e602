"""How the CPU baseline (reference SSE2 kernels under the oracle's populate) scales with host threads on this box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from octopus_amd import abi, synth
print("affinity", len(os.sched_getaffinity(0)), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a")
oracle.set_l1_backend("sse2" if oracle.have_ref() else "oracle")
cfg = abi.Config.default(max_indel_error=16)
batch = synth.batch_from_regions([synth.make_region(np.random.default_rng(42), 20000, 64, B=16, positions="none")])
for n in (1, 4, 16, 32, 64, 128, 256):
    t0 = time.perf_counter(); _, st, stats = oracle.populate(cfg, batch, n_threads=n); dt = time.perf_counter() - t0
    print(f"threads {n:4d}: {dt*1e3:8.1f} ms  {stats['band_cells']/dt/1e9:7.2f} GCUPS  {stats['band_cells']/dt/1e9/n:6.3f} per thread", flush=True)
