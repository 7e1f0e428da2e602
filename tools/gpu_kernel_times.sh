# Per-kernel time of one single-slice bench run (every launch has the device to itself), from rocprofv3's kernel trace:
#   bash tools/gpu_kernel_times.sh <tag> ["VAR=value" ...]      -> gpurun_out/<tag>/kernel_times.txt
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/${1:-ktimes}; mkdir -p $O; shift
env OCT_PHMM_SLICES=1 "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o b -- python bench.py --no-cpu-baseline --no-small-batch --steps 3 --warmup 1 --no-extras > $O/bench_prof.json 2> $O/err.log
python - "$O" <<'PY' | tee $O/kernel_times.txt
import sqlite3, glob, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + '/prof/*.db')[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1e6, sum(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 4 desc limit 16"
for name, calls, avg_ms, total_ms in db.execute(q):
    print(f"{name[:70]:70s} calls {calls:4d}  avg {avg_ms:8.3f} ms  total {total_ms:8.2f} ms")
PY
