# (this call hung: the rewritten between-flank loop of k_walk_rows it measured livelocked early-stopping walks, `timeout` without -k did not end the stuck python, and the call ran into gpurun's own limit - 25 GPU-minutes. The rewrite was reverted unmeasured)
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s37; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/rc.log
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace1 -o s -- python /root/repo/tools/multi_region_trace.py 1 > /root/repo/$O/trace1.log 2>&1)
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace4 -o s -- python /root/repo/tools/multi_region_trace.py 4 > /root/repo/$O/trace4.log 2>&1)
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace_cc -o s -- /root/repo/tools/region_calls_bench 300 300 24 1 1 > /root/repo/$O/trace_cc.log 2>&1)
cat $O/rc.log; tail -2 $O/pytest.log; grep "k_walk" $O/trace1/s_kernel_stats.csv $O/trace4/s_kernel_stats.csv $O/trace_cc/s_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
