# Per-kernel time of single-slice bench runs (every launch has the device to itself) for several workloads, from rocprofv3's kernel trace:
#   bash tools/gpu_kernel_split.sh <tag> <workload> [<workload> ...]     -> gpurun_out/<tag>/split_<workload>.txt + bench_<workload>.json
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/${1:-split}; mkdir -p $O; shift
for W in "$@"; do
  (cd /tmp && env OCT_PHMM_SLICES=1 timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_$W -o s -- \
     python /root/repo/bench.py --workload $W --no-cpu-baseline --no-small-batch --steps 3 --warmup 1 --no-extras > /root/repo/$O/bench_$W.json 2> /root/repo/$O/err_$W.log)
  echo "split $W rc=$?" >> $O/rc.log
  find $O/prof_$W -name "*kernel_trace.csv" -delete
  python - $O $W <<'PY' | tee $O/split_$W.txt
import csv, glob, json, sys
o, w = sys.argv[1], sys.argv[2]
f = glob.glob(f"{o}/prof_{w}/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
try:
    b = json.loads(open(f"{o}/bench_{w}.json").read().strip().splitlines()[-1])
    steps = b["steps"] + b["warmup"] + 4      # + the single-slice roofline leg's 1 + 3 runs
    s = b["stats"]
    print(f"## {w}: ms_per_step {b['ms_per_step']:.3f}  GCUPS {b['value']:.0f}  loglik/s {b['loglik_per_s']:.4g}  pairs {s['n_pairs']}  candidates {s['n_candidates']}  "
          f"fast_path {s['n_fast_path']}  dp {s['n_dp_score_only']}+{s['n_dp_traceback']}  shared pairs {s['n_pairs_shared']}  (kernel times: total ms / {steps} runs)")
except Exception as e:
    steps = 8; print("## no bench line:", e)
tot = 0.0
for r in rows[:22]:
    ms = float(r["TotalDurationNs"]) / 1e6; tot += ms
    print(f"{r['Name'][:86]:86s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e6:8.3f} ms  per run {ms/steps:8.3f} ms")
print(f"sum of listed kernels per run {tot/steps:.3f} ms")
PY
done
cat $O/rc.log
