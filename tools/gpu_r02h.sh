#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r02h; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python bench.py --no-cpu-baseline --no-small-batch --steps 5 --warmup 1 --no-extras > $O/bench.json 2> $O/err.log
python - <<'PY'
import sqlite3,glob
db=sqlite3.connect(glob.glob('gpurun_out/r02h/prof/*.db')[0])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
q=f"select s.kernel_name, count(*), avg(d.end-d.start)/1e6, sum(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 4 desc limit 14"
for r in db.execute(q): print(r[0][:60], r[1], round(r[2],3), round(r[3],2))
PY
