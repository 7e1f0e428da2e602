cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s26; mkdir -p $O
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --stats --output-format csv -d /root/repo/$O/api -o s -- /root/repo/tools/region_calls_bench 400 300 24 1 1 > /root/repo/$O/api.log 2>&1)
ls $O/api | head -20
head -30 $O/api/s_hip_api_stats.csv 2>/dev/null | cut -c1-160
python - <<'PY'
import csv,glob
base='/root/repo/gpurun_out/r03_s26/api/'
ev=[]
for f,kind in (('s_kernel_trace.csv','K'),('s_memory_copy_trace.csv','C'),('s_hip_api_trace.csv','A')):
    try: rows=list(csv.DictReader(open(base+f)))
    except Exception as e: print(f,e); continue
    for r in rows:
        name=r.get('Kernel_Name') or r.get('Direction') or r.get('Function') or r.get('Name')
        ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),kind,name))
ev.sort()
# find the last k_hap_tables and print events from 150us before to 350us after
ks=[e for e in ev if e[2]=='K' and 'k_hap_tables' in e[3]]
t0=ks[-5][0]
for s,e,k,n in ev:
    if t0-120000 <= s <= t0+300000:
        print(f"{(s-t0)/1e3:8.1f} {(e-s)/1e3:7.1f} {k} {n[:70]}")
PY
