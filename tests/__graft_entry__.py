"""Driver entry points: build() compiles everything (no GPU needed), smoke() runs one tiny populate on cuda:0."""
from __future__ import annotations

import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def build() -> None:
    """hipcc --offload-arch=gfx950 the HIP library in-tree, build the oracle (C restatement, plus the reference's
    own SIMD headers into oracle/_ref when /root/reference exists — building the checker is not using it)."""
    from octopus_amd import engine
    engine.build()
    import oracle
    oracle.build()
    import octopus_amd  # noqa: F401
    assert engine.LIB_PATH.exists()


def smoke() -> None:
    """One small invocation of the hot path on GPU 0 (populate: fast path + score-only DP + traceback DP + epilogue)
    checked against the CPU oracle."""
    import numpy as np
    from octopus_amd import abi, engine, synth
    import oracle

    batch = synth.config_batch("tiny", seed=7, B=16)
    cfg = abi.Config.default(max_indel_error=16)
    eng = engine.Engine(cfg)           # raises if liboct_phmm.so is missing or no gfx950 device is visible
    got, st = eng.populate(batch)
    want, wst, wstats = oracle.populate(cfg, batch)
    assert st.code == wst.code == abi.OK
    err = float(np.max(np.abs(got - want)))
    assert err <= 1e-9, f"GPU vs oracle max |delta ln L| = {err}"
    assert wstats["n_dp_score_only"] + wstats["n_dp_traceback"] > 0
    eng.close()
    print(f"smoke ok: {got.size} log-likelihoods, max |delta| = {err:.3g}, oracle stats {wstats}")


if __name__ == "__main__":
    build()
    if len(sys.argv) > 1 and sys.argv[1] == "smoke":
        smoke()
