#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r02m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "share or slices or 100k or stream" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.log
OCT_PHMM_SLICES=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python bench.py --no-cpu-baseline --no-small-batch --steps 3 --warmup 1 --no-extras > $O/bench_prof.json 2> $O/err.log
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; tail -3 $O/pytest.log
python - <<'PY'
import sqlite3,glob,json
db=sqlite3.connect(glob.glob('gpurun_out/r02m/prof/*.db')[0])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
q=f"select s.kernel_name, count(*), avg(d.end-d.start)/1e6, sum(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 4 desc limit 8"
for r in db.execute(q): print(r[0][:60], r[1], round(r[2],3), round(r[3],2))
d=json.loads(open("gpurun_out/r02m/bench.json").read().strip().splitlines()[-1])
s=d["stats"]; st=d.get("stream",{})
print("ms", round(d["ms_per_step"],2), "value", round(d["value"],1), "ref_work", round(d["gcups_reference_work"],1), "loglik/s", round(d["loglik_per_s"]/1e6,1),
      "shared pairs", s.get("n_pairs_shared"), "verified", d.get("verified_rows"), d.get("verified_max_abs_diff"),
      "| stream ms", st.get("ms"), "regions/s", st.get("regions_per_s"), "shared", st.get("pairs_shared"), "verified", st.get("verified_rows"), st.get("verified_max_abs_diff"),
      "e2e", d.get("e2e_ms_from_host"), "small", d.get("small_batch_ms"))
PY
