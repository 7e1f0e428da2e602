cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s19; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.log
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD"; do
  D=pmc_$(echo $C | cut -d' ' -f1)
  (cd /tmp && timeout 150 rocprofv3 --pmc $C --output-format csv -d /root/repo/$O/$D -o p -- python /root/repo/tools/multi_region_trace.py 1 > /root/repo/$O/$D.json 2> /root/repo/$O/$D.err); echo "$D rc=$?" >> $O/rc.log
done
for TH in 64 256; do
  (cd /tmp && OCT_PHMM_WALK_ROWS_THREADS=$TH timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace_th$TH -o s -- python /root/repo/tools/multi_region_trace.py 1 > /root/repo/$O/trace_th$TH.log 2>&1)
  echo "## rows threads $TH" >> $O/ab.log; grep "k_walk" $O/trace_th$TH/s_kernel_stats.csv | cut -c1-120 >> $O/ab.log
done
for I in 0 1; do
  echo "## int32 lanes $I (staged lockstep walk)" >> $O/ab.log
  OCT_PHMM_WALK_STAGE=1 OCT_LAT_INT32=$I OCT_LAT_SMALL=1 timeout 100 python tools/latency_breakdown.py 2>&1 | cut -c1-330 >> $O/ab.log
done
(cd /tmp && OCT_PHMM_WALK_STAGE=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace_bench -o s -- /root/repo/tools/region_calls_bench 300 300 24 1 1 > /root/repo/$O/trace_bench.log 2>&1)
(cd /tmp && OCT_PHMM_WALK_STAGE=2 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace_bench_rows -o s -- /root/repo/tools/region_calls_bench 300 300 24 1 1 > /root/repo/$O/trace_bench_rows.log 2>&1)
echo "## region_calls_bench regions: staged vs rows" >> $O/ab.log; grep "k_walk\|k_dp" $O/trace_bench/s_kernel_stats.csv $O/trace_bench_rows/s_kernel_stats.csv | cut -c1-160 >> $O/ab.log
cat $O/rc.log; tail -3 $O/pytest.log; cat $O/ab.log
python - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('/root/repo/gpurun_out/r03_s19/pmc_*/')):
    for f in glob.glob(d+'*counter_collection.csv'):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if 'k_walk_rows' in r['Kernel_Name'] or 'k_dp<16, true' in r['Kernel_Name']:
                acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items():
            print(k, {c: sum(x)/len(x) for c,x in v.items()})
PY
