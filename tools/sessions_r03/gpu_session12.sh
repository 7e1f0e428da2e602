cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_parity.log 2>&1; echo "pytest rc=$?" >> $O/rc.log
timeout 100 python tools/latency_breakdown.py > $O/latency.json 2>&1
timeout 200 ./tools/region_calls_bench 2000 300 24 1 16 > $O/region_calls.log 2>&1
OCT_PHMM_SLICES=1 timeout 300 bash -c 'cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/'$O'/kstats -o s -- python /root/repo/bench.py --no-small-batch --no-cpu-baseline --no-extras --steps 3 --warmup 1 > /root/repo/'$O'/bench_1slice.json 2> /root/repo/'$O'/kstats.err'; echo "kstats rc=$?" >> $O/rc.log
find $O -name "*kernel_trace.csv" -delete
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; tail -3 $O/pytest_parity.log; cut -c1-420 $O/latency.json; cat $O/region_calls.log | cut -c1-200; head -14 $O/kstats/s_kernel_stats.csv | cut -c1-150
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_s12/bench.json'))
for k in ('value','ms_per_step','verified_max_abs_diff','e2e_ms_from_host','region_call_ms','region_server_regions_per_s','small_batch_ms'):
    print(k, d.get(k))
print('stream', d.get('stream'))
print('long_read', d.get('long_read'))
print('valu', {k:v for k,v in d['roofline']['valu'].items() if k!='note'})
PY
