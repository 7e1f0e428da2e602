cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s27; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.log
for R in 1 2 3; do for M in 1 0; do
  echo "## rep $R host_mapped=$M" >> $O/region_calls.log
  OCT_PHMM_HOST_MAPPED=$M timeout 200 ./tools/region_calls_bench 3000 300 24 1 16 2>&1 | grep -v "threads\": [48]" | cut -c1-150 >> $O/region_calls.log
done; done
OCT_LAT_SMALL=1 timeout 100 python tools/latency_breakdown.py > $O/latency.json 2>&1
cat $O/rc.log; tail -3 $O/pytest.log; cat $O/region_calls.log; cut -c1-400 $O/latency.json
