cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "staged or populate or random or late or align or server" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.log
timeout 100 python tools/latency_breakdown.py > $O/latency.json 2>&1
timeout 100 python tools/multi_region_trace.py 1 > $O/mr1.json 2>&1; timeout 100 python tools/multi_region_trace.py 4 > $O/mr4.json 2>&1
timeout 200 ./tools/region_calls_bench 2000 300 24 1 16 > $O/region_calls.log 2>&1
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/region_trace -o s -- python /root/repo/tools/multi_region_trace.py 1 > /root/repo/$O/region_trace.log 2>&1)
cat $O/rc.log; tail -2 $O/pytest.log; cut -c1-300 $O/latency.json; cut -c1-200 $O/mr1.json $O/mr4.json; cat $O/region_calls.log | cut -c1-180; grep k_walk $O/region_trace/s_kernel_stats.csv | cut -c1-140
