# Round 6, session 36 (the final kernels, four slices): do the slices of the 12.8 M-pair step overlap on the device? Kernel trace (start / end / queue of every dispatch) of three default steps.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s36; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/trace -o t -- python /root/repo/bench.py --no-small-batch --no-cpu-baseline --no-extras --steps 3 --warmup 1 > /root/repo/$O/bench.json 2> /root/repo/$O/bench.err)
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python - "$f" > $O/overlap.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
ev.sort()
t0 = ev[0][0]
# the last step = the last third of the dispatches that are not the clock probe
main = [e for e in ev if "clock_probe" not in e[2]]
print("dispatches", len(main), "queues", sorted(set(e[3] for e in main)))
def short(n):
    n = n.replace("octphmm::", "").replace("void ", "")
    return n[:n.index("(")] if "(" in n else n[:40]
# find step boundaries: gaps > 1 ms between consecutive kernels' activity
busy_end = main[0][1]; steps = [[main[0]]]
for e in main[1:]:
    if e[0] - busy_end > 300000: steps.append([])
    steps[-1].append(e); busy_end = max(busy_end, e[1])
print("activity blocks:", [len(s) for s in steps])
for s in [x for x in steps if len(x) > 100]:
    a = min(e[0] for e in s); z = max(e[1] for e in s)
    tot = sum(e[1] - e[0] for e in s)
    dp = sum(e[1] - e[0] for e in s if "k_dp" in e[2])
    # union coverage of DP kernels, and time where NO dp kernel runs
    iv = sorted((e[0], e[1]) for e in s if "k_dp" in e[2]); cov = 0; cur_a, cur_b = iv[0]
    for x, y in iv[1:]:
        if x > cur_b: cov += cur_b - cur_a; cur_a, cur_b = x, y
        else: cur_b = max(cur_b, y)
    cov += cur_b - cur_a
    print(f"block span {(z-a)/1e6:.3f} ms  sum of durations {tot/1e6:.3f}  dp sum {dp/1e6:.3f}  dp union {cov/1e6:.3f}  span without any dp kernel {(z-a-cov)/1e6:.3f}")
    for e in s[-125:]:
        print(f"{(e[0]-a)/1e3:10.1f} us  dur {(e[1]-e[0])/1e3:9.1f}  q{e[3]}  {short(e[2])}")
PY
head -5 $O/overlap.txt | cut -c1-150
rm -rf $O/trace
