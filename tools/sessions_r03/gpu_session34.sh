cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s34; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.log
for R in 1 2 3; do
  timeout 200 ./tools/region_calls_bench 3000 300 24 1 8 16 2>&1 | grep -v "plain calls" | cut -c1-150 >> $O/region_calls.log
done
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace4 -o s -- python /root/repo/tools/multi_region_trace.py 4 > /root/repo/$O/trace4.log 2>&1)
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/r03_s34/trace4/s_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_hap_tables' in r['Kernel_Name']][-1]
t0=int(rows[idx]['Start_Timestamp'])
for r in rows[idx:idx+16]:
    print(f"  {(int(r['Start_Timestamp'])-t0)/1e3:7.1f} us dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:6.1f}  {r['Kernel_Name'][:60]}")
PY
cat $O/rc.log; tail -2 $O/pytest.log; cat $O/region_calls.log
