cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s21; mkdir -p $O
WORKERS=3 bash tools/gpu_server_profile.sh r04_s21 64 128 > /dev/null 2>&1
(cd /tmp && OCT_BENCH_REPS=2 timeout -k 5 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /root/repo/$O/trace -o s -- /root/repo/tools/region_calls_bench --file /tmp/stream_regions.bin 64 > /root/repo/$O/trace.log 2> /root/repo/$O/trace.err)
python - <<'PY' > $O/busy.txt
import csv, glob, collections
rows = []
for f in glob.glob("/root/repo/gpurun_out/r04_s21/trace/**/s_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("octphmm::", "").replace("void ", "")[:40]))
rows.sort()
# the second half of the trace (steady state)
t_lo = rows[len(rows) // 2][0]; rows = [r for r in rows if r[0] >= t_lo]
wall = rows[-1][1] - rows[0][0]
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]; depth_time = collections.Counter()
for s, e, _ in rows[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in rows: tot[n] += e - s; cnt[n] += 1
print(f"steady-state window {wall / 1e6:.1f} ms, {len(rows)} launches; some kernel running {busy / wall:.3f} of the time; sum of kernel durations {sum(tot.values()) / wall:.3f} x wall")
# concurrency histogram
ev = sorted([(s, 1) for s, e, _ in rows] + [(e, -1) for s, e, _ in rows]); d = 0; last = ev[0][0]
for t, k in ev:
    depth_time[d] += t - last; last = t; d += k
print("time share by number of kernels in flight:", {k: round(v / wall, 3) for k, v in sorted(depth_time.items())})
for n, v in tot.most_common(14): print(f"  {n:42s} {cnt[n]:6d} launches  {v / wall:.3f} x wall   avg {v / cnt[n] / 1e3:7.1f} us")
PY
find $O -name "*.csv" -size +1M -delete
cat $O/server_profile.log $O/busy.txt; tail -3 $O/trace.log | cut -c1-300
