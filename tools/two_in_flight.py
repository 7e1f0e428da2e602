"""Round 6, session 31: the headline batch with TWO steps in flight (two handles, two resident images of the same 100k x 128 region, step i + 1 enqueued before step i is waited for)
against the bench's serial run-wait loop: what a caller with several batches in flight gets from the same kernels (the front end of one step beside the tail of the other)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from octopus_amd import abi, engine, synth

B = 16
workload = sys.argv[1] if len(sys.argv) > 1 else "100kx128"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = abi.Config.default(max_indel_error=B)
if workload.startswith("stream"):
    regions = synth.region_stream_shard(seed=42, n_regions=2000, rank=0, world=1, B=B, positions="none", workers=8, hq=workload == "stream-hq")
else:
    regions = [synth.config_region(workload, seed=42, B=B, positions="none")]
batch = synth.batch_from_regions(regions)
engs = [engine.Engine(cfg) for _ in range(3)]
rbs = [e.upload(batch) for e in engs]
for rb in rbs: rb.run(); rb.wait()
ref = rbs[0].download().copy()
out = {"workload": workload, "steps": steps}
for depth in (1, 2, 3, 1, 2, 3):
    for rb in rbs[:depth]: rb.run(); rb.wait()
    t0 = time.perf_counter()
    inflight = []
    for i in range(steps):
        rb = rbs[i % depth]
        if len(inflight) == depth: inflight.pop(0).wait()
        rb.run(); inflight.append(rb)
    for rb in inflight: rb.wait()
    dt = (time.perf_counter() - t0) / steps
    out.setdefault(f"ms_per_step_{depth}_in_flight", []).append(round(dt * 1e3, 3))
st = rbs[0].stats()
cells = st["band_cells"] - st.get("band_cells_shared", 0)
for depth in (1, 2, 3):
    out[f"gcups_{depth}_in_flight"] = round(cells / (min(out[f"ms_per_step_{depth}_in_flight"]) * 1e-3) / 1e9, 1)
out["results_equal"] = all(bool(np.array_equal(rb.download(), ref)) for rb in rbs)
print(json.dumps(out))
