"""BASELINE.json configs[4] at full size on the GPU: 64 x 10 kb reads against 8 x 20 kb haplotypes, band 256, int32 lanes,
device k-mer mapping; checked against the oracle's upper layers driving the reference's own SIMD kernels (AVX-512/AVX2/SSE2 int32,
band 256) on the host."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from octopus_amd import abi, engine, synth
# default: BASELINE configs[4] (band 256). "ccs": the reference's PacBioCCS.config shape - band 16, int32 lanes, 10 kb reads x 14 kb haplotypes
if len(sys.argv) > 1 and sys.argv[1] == "ccs":
    rng = np.random.default_rng(42)
    t0 = time.time(); batch = synth.batch_from_regions([synth.make_region(rng, 256, 12, T=10_000, Lh=14_000, B=16, flank=(400, 400), positions="none",
                                                                          q_values=(20, 40), indels_per_read=4)])
    cfg = abi.Config.default(max_indel_error=16, use_int_scores=1)
else:
    t0 = time.time(); batch = synth.config_batch("long64x8", seed=42, B=256, positions="none")
    cfg = abi.Config.default(max_indel_error=256, use_int_scores=1)
print("generated in %.1fs" % (time.time() - t0), flush=True)
eng = engine.Engine(cfg)
rb = eng.upload(batch)
rb.run(); got = rb.download().copy()
times = []
for _ in range(3):
    t0 = time.perf_counter(); rb.run(); rb.wait(); times.append(time.perf_counter() - t0)
stats = rb.stats(); dp_ms, n = rb.kernel_time()
print(json.dumps(dict(ms=min(times) * 1e3, gcups=stats["band_cells"] / min(times) / 1e9, dp_ms=dp_ms, launches=n, stats=stats)), flush=True)
oracle.set_l1_backend("native" if oracle.have_ref() else "oracle")
t0 = time.time(); want, st, wstats = oracle.populate(cfg, batch, n_threads=oracle.host_cores()); dt = time.time() - t0
print("oracle(reference kernels) %.1fs on %d threads: %.2f GCUPS" % (dt, oracle.host_cores(), wstats["band_cells"] / dt / 1e9), flush=True)
assert st.code == 0 and wstats == stats, (wstats, stats)
print("max |delta| =", float(np.max(np.abs(got - want))), "n =", got.size)
assert np.max(np.abs(got - want)) <= 1e-9
