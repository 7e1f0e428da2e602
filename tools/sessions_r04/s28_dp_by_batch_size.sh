cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s28; mkdir -p $O
for N in 4 16 64 256; do
  (cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/k_$N -o s -- python /root/repo/tools/mid_batch_trace.py $N > /root/repo/$O/k_$N.json 2> /root/repo/$O/k_$N.err)
  python - $N <<'PY' >> $O/dp.txt
import csv, sys, json
n = sys.argv[1]
st = json.loads(open(f"/root/repo/gpurun_out/r04_s28/k_{n}.json").read().strip().splitlines()[-1])["stats"]
print(f"## {n} regions: pairs {st['n_pairs']} score-only tasks {st['n_dp_score_only']} traceback tasks {st['n_dp_traceback']} shared pairs {st['n_pairs_shared']} band cells {st['band_cells']}")
ideal = (st['n_dp_score_only'] - st.get('n_dp_score_only_shared', 0)) * 166 * 103 / 8 / (1024 * 2.4e9) * 1e6, (st['n_dp_traceback'] - st.get('n_dp_traceback_shared', 0)) * 166 * 181 / 8 / (1024 * 2.4e9) * 1e6
print(f"   ideal at the flat batch's cost per iteration: score-only {ideal[0]:.0f} us, traceback {ideal[1]:.0f} us")
for r in csv.DictReader(open(f"/root/repo/gpurun_out/r04_s28/k_{n}/s_kernel_stats.csv")):
    if "k_dp" in r["Name"] or "k_walk" in r["Name"] or "k_kmer_map" in r["Name"]:
        print(f"   {r['Name'].split('(')[0].replace('void octphmm::', '').replace('octphmm::', ''):36s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:8.1f}")
PY
  find $O -name "*kernel_trace.csv" -delete
done
cat $O/dp.txt
