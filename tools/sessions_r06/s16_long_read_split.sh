# Round 6, session 16: kernel split of the unsplit long-read legs (ccs256x12, ccs2048x12: band 16, int32 lanes: k_dp_rows + k_walk_rows) - where their time goes.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s16; mkdir -p $O
for L in ccs256x12 ccs2048x12; do
  (cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/k_$L -o s -- python /root/repo/tools/long_read_legs.py $L > /root/repo/$O/leg_$L.json 2> /root/repo/$O/leg_$L.err)
  find $O/k_$L -name "*kernel_trace.csv" -delete
  echo "## $L: $(cut -c1-200 $O/leg_$L.json | tail -1)"
  python - $(find $O/k_$L -name "*kernel_stats.csv") <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e6:8.3f} ms  total/4 runs {float(r['TotalDurationNs'])/4e6:8.3f} ms")
PY
done
