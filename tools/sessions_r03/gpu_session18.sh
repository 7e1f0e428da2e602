cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s18; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_l1.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.log
for M in 2 1; do for F in 0 1; do
  echo "## WALK_STAGE=$M FORK_EARLY=$F" >> $O/ab.log
  OCT_PHMM_WALK_STAGE=$M OCT_PHMM_DSL_FORK_EARLY=$F timeout 100 python tools/latency_breakdown.py 2>&1 | cut -c1-330 >> $O/ab.log
  OCT_PHMM_WALK_STAGE=$M OCT_PHMM_DSL_FORK_EARLY=$F timeout 200 ./tools/region_calls_bench 2000 300 24 1 16 2>&1 | grep -v "threads\": [48]" | cut -c1-170 >> $O/ab.log
done; done
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/region_trace -o s -- python /root/repo/tools/multi_region_trace.py 1 > /root/repo/$O/region_trace.log 2>&1)
(cd /tmp && OCT_PHMM_DSL_FORK_EARLY=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/region_trace_fe -o s -- python /root/repo/tools/multi_region_trace.py 1 > /root/repo/$O/region_trace_fe.log 2>&1)
cat $O/rc.log; tail -3 $O/pytest.log; cat $O/ab.log; grep "k_walk\|k_dp" $O/region_trace/s_kernel_stats.csv | cut -c1-140
