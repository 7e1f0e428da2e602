#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r02l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "share or slices or 100k or stream" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.log
for sl in 3 4 6 8; do
    OCT_PHMM_SLICES=$sl timeout 300 python bench.py --no-cpu-baseline --no-small-batch --steps 10 --warmup 2 --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slices $sl ms', round(d['ms_per_step'],2), 'shared', d['stats']['n_pairs_shared'], 'loglik/s', round(d['loglik_per_s']/1e6,1))"
done | tee $O/slices.txt
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; tail -4 $O/pytest.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02l/bench.json").read().strip().splitlines()[-1])
s=d["stats"]; st=d.get("stream",{})
print("ms", round(d["ms_per_step"],2), "value", round(d["value"],1), "ref_work", round(d["gcups_reference_work"],1), "loglik/s", round(d["loglik_per_s"]/1e6,1),
      "shared pairs", s.get("n_pairs_shared"), "verified", d.get("verified_rows"), d.get("verified_max_abs_diff"),
      "| stream ms", st.get("ms"), "regions/s", st.get("regions_per_s"), "shared", st.get("pairs_shared"), "verified", st.get("verified_rows"), st.get("verified_max_abs_diff"),
      "e2e", d.get("e2e_ms_from_host"), "small", d.get("small_batch_ms"))
PY
