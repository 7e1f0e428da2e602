#!/bin/bash
# round 2, step 5: all GPU tests with pair de-duplication in the default path, then the bench line with and without it
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r02k; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" > $O/rc.log
OCT_PHMM_DEDUP=0 timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_dedup0.json 2> $O/bench_dedup0.err; echo "bench0 rc=$?" >> $O/rc.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; tail -9 $O/pytest_gpu.log
python - <<'PY'
import json
for m in ("bench_dedup0","bench"):
    d=json.loads(open(f"gpurun_out/r02k/{m}.json").read().strip().splitlines()[-1])
    s=d["stats"]; st=d.get("stream",{})
    print(m, "ms", round(d["ms_per_step"],2), "value", round(d["value"],1), "ref_work", round(d["gcups_reference_work"],1), "loglik/s", round(d["loglik_per_s"]/1e6,1),
          "shared pairs", s.get("n_pairs_shared"), "verified", d.get("verified_rows"), d.get("verified_max_abs_diff"),
          "| stream ms", st.get("ms"), "regions/s", st.get("regions_per_s"), "shared", st.get("pairs_shared"), "verified", st.get("verified_rows"), st.get("verified_max_abs_diff"),
          "e2e", d.get("e2e_ms_from_host"), "small", d.get("small_batch_ms"), "long", d.get("long_read",{}).get("ms"), "vs", d.get("vs_baseline"))
PY
