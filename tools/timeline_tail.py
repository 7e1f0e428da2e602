#!/usr/bin/env python3
"""Kernel / copy timeline of the LAST burst of device work in a rocprofv3 --kernel-trace [--memory-copy-trace] --output-format csv directory (what
tools/mid_batch_trace.py's marker populate left):  python tools/timeline_tail.py <trace dir> [gap_ms=20]
Prints start, duration, gap to the previous entry's end and the stream / queue per entry, then span, summed durations and the number of kernel launches."""
import csv, glob, sys
d = sys.argv[1]
gap_ns = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 20_000_000
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("octphmm::", "").replace("void ", "")[:48], r.get("Stream_Id", r.get("Queue_Id", "")), 1))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), "", 0))
rows.sort()
if not rows:
    print("no trace rows under", d); sys.exit(1)
i = len(rows) - 1
while i > 0 and rows[i][0] - rows[i - 1][1] < gap_ns: i -= 1
t0 = rows[i][0]; prev = t0; busy = 0; launches = 0
for s, e, name, q, is_kernel in rows[i:]:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:7.1f}  {q:>4} {name}")
    prev = max(prev, e); busy += e - s; launches += is_kernel
print(f"span {(prev - t0) / 1e3:.1f} us, sum of durations {busy / 1e3:.1f} us, {launches} kernel launches, {len(rows) - i - launches} copies")
