cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "device_sized or staged or populate_basic or templates or random_scenarios or late_traceback or server or chunked or empty or align" > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/rc.log
g++ -O2 -std=c++17 tools/region_calls_bench.cpp -o tools/region_calls_bench -Iinclude -Loctopus_amd -loct_phmm -Wl,-rpath,/root/repo/octopus_amd -lpthread 2>&1 | tail -3
for N in 1 4 8; do
  for V in "A=1" "OCT_PHMM_DEVICE_SIZED=0"; do
    echo "## regions=$N $V" >> $O/multi_region.log
    env $V timeout 100 python tools/multi_region_trace.py $N >> $O/multi_region.log 2>&1
  done
done
for V in "A=1" "OCT_PHMM_DEVICE_SIZED=0" "A=2"; do
  echo "## $V" >> $O/server_profile.log
  env $V OCT_PHMM_SERVER_PROFILE=1 timeout 200 ./tools/region_calls_bench 2000 300 24 1 8 16 32 2>&1 | tail -12 >> $O/server_profile.log
done
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace4 -o s -- python /root/repo/tools/multi_region_trace.py 4 > /root/repo/$O/trace4.log 2>&1); echo "trace4 rc=$?" >> $O/rc.log
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace1 -o s -- python /root/repo/tools/multi_region_trace.py 1 > /root/repo/$O/trace1.log 2>&1); echo "trace1 rc=$?" >> $O/rc.log
timeout 100 python tools/latency_breakdown.py > $O/latency.json 2>&1
cat $O/rc.log; tail -3 $O/pytest_subset.log; cat $O/multi_region.log | cut -c1-200; cat $O/server_profile.log | cut -c1-300; cut -c1-420 $O/latency.json
