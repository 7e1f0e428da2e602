#!/bin/bash
# round 2, step 5: exact de-duplication of pairs - parity on the GPU, then A/B of the bench batch and the stream with and without it
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r02g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "share or late or slices or random" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.log
for mode in 0 1; do
  OCT_PHMM_DEDUP=$mode timeout 600 python bench.py --no-cpu-baseline --no-small-batch --steps 10 --warmup 2 > $O/bench_dedup$mode.json 2> $O/bench_dedup$mode.err; echo "bench$mode rc=$?" >> $O/rc.log
done
timeout 600 python bench.py --no-cpu-baseline --no-small-batch --steps 10 --warmup 2 --no-extras > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?" >> $O/rc.log
cat $O/rc.log; tail -5 $O/pytest.log
python - <<'PY'
import json
for m in ("dedup0","dedup1","default"):
    try:
        d=json.loads(open(f"gpurun_out/r02g/bench_{m}.json").read().strip().splitlines()[-1])
        s=d["stats"]; st=d.get("stream",{})
        print(m, "ms", round(d["ms_per_step"],2), "value", round(d["value"],1), "ref_work", round(d["gcups_reference_work"],1), "loglik/s", round(d["loglik_per_s"]/1e6,1),
              "shared pairs", s.get("n_pairs_shared"), "verified", d.get("verified_rows"), d.get("verified_max_abs_diff"),
              "| stream ms", st.get("ms"), "regions/s", st.get("regions_per_s"), "shared", st.get("pairs_shared"), "verified", st.get("verified_rows"), st.get("verified_max_abs_diff"), "e2e", d.get("e2e_ms_from_host"))
    except Exception as e: print(m, "ERR", e)
PY
