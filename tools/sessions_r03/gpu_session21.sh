cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s21; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.log
for F in 0 1; do
  (cd /tmp && OCT_PHMM_DSL_FORK_EARLY=$F timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace_py_f$F -o s -- python /root/repo/tools/multi_region_trace.py 1 > /root/repo/$O/trace_py_f$F.log 2>&1)
  (cd /tmp && OCT_PHMM_DSL_FORK_EARLY=$F timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace_cc_f$F -o s -- /root/repo/tools/region_calls_bench 300 300 24 1 1 > /root/repo/$O/trace_cc_f$F.log 2>&1)
  echo "## fork early $F: python region / C++ bench regions" >> $O/ab.log
  grep "k_walk\|k_dp" $O/trace_py_f$F/s_kernel_stats.csv $O/trace_cc_f$F/s_kernel_stats.csv | cut -d, -f1-4 | cut -c20-150 >> $O/ab.log
  OCT_PHMM_DSL_FORK_EARLY=$F OCT_LAT_SMALL=1 timeout 100 python tools/latency_breakdown.py 2>&1 | cut -c1-330 >> $O/ab.log
  OCT_PHMM_DSL_FORK_EARLY=$F timeout 200 ./tools/region_calls_bench 2000 300 24 1 16 2>&1 | grep -v "threads\": [48]" | cut -c1-170 >> $O/ab.log
done
cat $O/rc.log; tail -3 $O/pytest.log; cat $O/ab.log
python - <<'PY'
import csv
for d in ('trace_py_f0','trace_py_f1','trace_cc_f1'):
    rows=list(csv.DictReader(open(f'/root/repo/gpurun_out/r03_s21/{d}/s_kernel_trace.csv')))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    idx=[i for i,r in enumerate(rows) if 'k_hap_tables' in r['Kernel_Name']][-1]
    t0=int(rows[idx]['Start_Timestamp'])
    print(d)
    for r in rows[idx:idx+12]:
        print(f"  {(int(r['Start_Timestamp'])-t0)/1e3:7.1f} us dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:6.1f}  {r['Kernel_Name'][:60]}")
PY
