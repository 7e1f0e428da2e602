#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r02n; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "share or slices" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.log
OCT_PHMM_SLICES=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python bench.py --no-cpu-baseline --no-small-batch --steps 3 --warmup 1 --no-extras > $O/bench_prof.json 2> $O/err.log
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --no-extras > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; tail -3 $O/pytest.log
python - <<'PY'
import sqlite3,glob,json
db=sqlite3.connect(glob.glob('gpurun_out/r02n/prof/*.db')[0])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
q=f"select s.kernel_name, count(*), avg(d.end-d.start)/1e6, sum(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 4 desc limit 12"
for r in db.execute(q): print(r[0][:60], r[1], round(r[2],3), round(r[3],2))
d=json.loads(open("gpurun_out/r02n/bench.json").read().strip().splitlines()[-1])
print("ms", round(d["ms_per_step"],2), "loglik/s", round(d["loglik_per_s"]/1e6,1), "shared", d["stats"]["n_pairs_shared"], "small", d.get("small_batch_ms"))
PY
