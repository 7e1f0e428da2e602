cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s26; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
(cd /tmp && OCT_PHMM_SERVER_WORKERS=${WORKERS:-3} OCT_BENCH_REPS=2 timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/trace -o s -- /root/repo/tools/region_calls_bench --file /tmp/stream_regions.bin ${CALLERS:-64} > /root/repo/$O/trace.log 2> /root/repo/$O/trace.err)
python - <<'PY' > $O/cycles.txt
import csv, glob, collections, statistics as st
rows = []
for f in glob.glob("/root/repo/gpurun_out/r04_s26/trace/**/s_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("octphmm::", "").replace("void ", "")[:40], r["Thread_Id"], int(r["Grid_Size_X"])))
rows.sort()
t_lo = rows[len(rows) // 2][0]; t_hi = rows[-1][1] - 20_000_000
by = collections.defaultdict(list)
for r in rows:
    if t_lo <= r[0] <= t_hi: by[r[3]].append(r)
print("threads launching kernels in the steady-state window:", {k: len(v) for k, v in by.items()})
for th, rs in by.items():
    if len(rs) < 200: continue
    starts = [i for i, r in enumerate(rs) if r[2] == "k_hap_tables"]
    spans, gaps, pro, dp, tail, nk, pairs = [], [], [], [], [], [], []
    for a, b in zip(starts, starts[1:]):
        seg = rs[a:b]
        end = max(r[1] for r in seg)
        spans.append((end - seg[0][0]) / 1e3); gaps.append((rs[b][0] - end) / 1e3); nk.append(len(seg))
        first_dp = next((r for r in seg if r[2].startswith("k_dp")), None)
        if first_dp: pro.append((first_dp[0] - seg[0][0]) / 1e3)
        cl = next((r for r in seg if r[2] == "k_classify"), None)
        if cl: pairs.append(cl[4])
    q = lambda v: (round(st.median(v), 1), round(st.mean(v), 1), round(max(v), 1)) if v else None
    print(f"worker thread {th}: {len(spans)} device batches; device span (k_hap_tables -> last kernel end) median/mean/max us {q(spans)}; turnaround to the next batch's first kernel {q(gaps)}; "
          f"first kernel -> first DP {q(pro)}; launches per batch {q(nk)}; classify grid (threads ~ pairs) {q(pairs)}")
    tot = sum(spans) + sum(gaps)
    print(f"   share of the worker's time: device span {sum(spans) / tot:.2f}, between batches {sum(gaps) / tot:.2f}")
PY
find $O -name "*.csv" -size +1M -delete
cat $O/cycles.txt; tail -2 $O/trace.log | cut -c1-250
