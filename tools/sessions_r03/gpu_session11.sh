cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s11; mkdir -p $O
python -c "from octopus_amd import engine; print(engine.kernel_source_sha())" > $O/kernel_source_sha
timeout 1200 python -m pytest tests -x -q -m gpu --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; tail -18 $O/pytest_gpu.log; tail -2 $O/smoke.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_s11/bench.json'))
for k in ('value','ms_per_step','vs_baseline','verified_rows','verified_max_abs_diff','e2e_ms_from_host','region_call_ms','region_server_regions_per_s','small_batch_ms'):
    print(k, d.get(k))
print('stream', d.get('stream'))
print('long_read', d.get('long_read'))
print('region_calls', json.dumps(d.get('region_calls'))[:900])
print('valu', {k:v for k,v in d['roofline']['valu'].items() if k!='note'})
print('roofline', {k:v for k,v in d['roofline'].items() if k not in ('valu','traffic_note','kernel')})
print('cpu', d.get('cpu_baseline'))
PY
tail -5 $O/bench.err
