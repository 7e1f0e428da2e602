cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s33; mkdir -p $O
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace4 -o s -- python /root/repo/tools/multi_region_trace.py 4 > /root/repo/$O/trace4.log 2>&1)
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace8 -o s -- python /root/repo/tools/multi_region_trace.py 8 > /root/repo/$O/trace8.log 2>&1)
python - <<'PY'
import csv
for d in ('trace4','trace8'):
    rows=list(csv.DictReader(open(f'/root/repo/gpurun_out/r03_s33/{d}/s_kernel_trace.csv')))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    idx=[i for i,r in enumerate(rows) if 'k_hap_tables' in r['Kernel_Name']][-1]
    t0=int(rows[idx]['Start_Timestamp'])
    print(d)
    for r in rows[idx:idx+13]:
        print(f"  {(int(r['Start_Timestamp'])-t0)/1e3:7.1f} us dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:6.1f}  {r['Kernel_Name'][:60]}")
PY
cat $O/trace4.log | cut -c1-200
