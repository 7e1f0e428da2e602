#!/usr/bin/env python3
"""The long-read legs of bench.py on their own (resident timing only; bench.py and tests/test_gpu_fullsize.py verify them): long64x8 / long512x8 (configs[4] and 8 x its reads,
band 256, int32), ccs-linked (PacBioCCS.config: 500-base linked chunks, band 16, int16), ccs256x12 (unsplit 10-14 kb reads at band 16, int32), ccs2048x12 (eight times its reads).   python tools/long_read_legs.py [leg ...]"""
import json
import os
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from octopus_amd import abi, engine, synth   # noqa: E402

legs = sys.argv[1:] or ["long64x8", "long512x8", "ccs-linked", "ccs256x12", "ccs2048x12"]
for leg in legs:
    if leg in ("long64x8", "long512x8"):
        cfg, regs = abi.Config.default(max_indel_error=256, use_int_scores=1), [synth.config_region(leg, seed=42, B=256, positions="none")]
    elif leg == "ccs-linked":
        cfg, regs = abi.Config.default(max_indel_error=16), synth.linked_stream(42, 1000, B=16)
    else:
        cfg, regs = abi.Config.default(max_indel_error=16, use_int_scores=1, use_mapping_quality=0), [synth.config_region(leg if leg.startswith("ccs") else "ccs256x12", seed=42, B=16, positions="none")]
    eng = engine.Engine(cfg)
    eng.set_timing(True)
    rb = eng.upload(synth.batch_from_regions(regs))
    rb.run(); rb.wait()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); rb.run(); rb.wait(); ts.append(time.perf_counter() - t0)
    if os.environ.get("OCT_TRACE_MARK"):      # a last step on its own, for tools/timeline_tail.py
        time.sleep(0.05); rb.run(); rb.wait()
    st = rb.stats()
    print(json.dumps({"leg": leg, "ms": min(ts) * 1e3, "gcups": (st["band_cells"] - st.get("band_cells_shared", 0)) / min(ts) / 1e9, "dp_kernel_ms_by_kind": rb.kernel_time_by_kind(),
                      "device_sized": rb.device_sized(), "stats": st}))
    rb.free(); eng.close()
