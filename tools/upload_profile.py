import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from octopus_amd import abi, engine, synth
cfg = abi.Config.default(max_indel_error=16)
batch = synth.batch_from_regions(synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
eng = engine.Engine(cfg)
pool = engine.PinnedPool(); locked = pool.batch(batch); out = pool.empty(batch.out_size(), np.float64)
for bt in (batch, locked):
    for _ in range(3):
        t0 = time.perf_counter(); eng.populate(bt, out=out); print("populate ms", (time.perf_counter() - t0) * 1e3, file=sys.stderr)
