# Round 6, session 17: k_kmer_map_big with staged tables, 16-bit counters and merged votes: the long-read legs, kernel split, the long-read / big-mapper GPU tests.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "big or 40k" > $O/gpu_tests_subset.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests_subset.log
for V in default big_nostage; do L=""; [ $V != default ] && L="OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_$V.so"; echo "## $V"; env $L timeout 300 python tools/long_read_legs.py ccs256x12 ccs2048x12 long64x8 2>/dev/null | cut -c1-110; done
for L in ccs2048x12; do
  (cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/k_$L -o s -- python /root/repo/tools/long_read_legs.py $L > /root/repo/$O/leg_$L.json 2> /root/repo/$O/leg_$L.err)
  find $O/k_$L -name "*kernel_trace.csv" -delete
  python - $(find $O/k_$L -name "*kernel_stats.csv") <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e6:8.3f} ms")
PY
done
