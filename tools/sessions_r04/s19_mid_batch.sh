cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s19; mkdir -p $O
for N in 16 64; do
  timeout -k 5 200 python tools/mid_batch_trace.py $N > $O/stages_$N.json 2> $O/stages_$N.err
  (cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /root/repo/$O/trace_$N -o s -- python /root/repo/tools/mid_batch_trace.py $N > /root/repo/$O/trace_$N.json 2> /root/repo/$O/trace_$N.err)
  python - $N <<'PY' > $O/timeline_$N.txt
import csv, sys, glob
n = sys.argv[1]
d = f"/root/repo/gpurun_out/r04_s19/trace_{n}"
rows = []
for f in glob.glob(d + "/**/s_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("octphmm::", "").replace("void ", "")[:44], r.get("Stream_Id", r.get("Queue_Id", ""))))
for f in glob.glob(d + "/**/s_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), ""))
rows.sort()
# the last populate: walk back from the end to the last gap > 20 ms
last = len(rows) - 1; i = last
while i > 0 and rows[i][0] - rows[i - 1][1] < 20_000_000: i -= 1
t0 = rows[i][0]; prev = t0; busy = 0
for s, e, name, q in rows[i:]:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:7.1f}  {q:>4} {name}")
    prev = max(prev, e); busy += e - s
print(f"span {(prev - t0) / 1e3:.1f} us, sum of durations {busy / 1e3:.1f} us, {len(rows) - i} entries")
PY
  find $O -name "*.csv" -size +2M -delete
done
cat $O/stages_*.json; tail -3 $O/timeline_16.txt $O/timeline_64.txt
