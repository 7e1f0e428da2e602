cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s34; mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py -x -q -k "narrow or long or wide or int32 or align or basic" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.log
timeout -k 5 300 python tools/long_read_legs.py ccs256x12 ccs2048x12 > $O/legs.json 2> $O/legs.err; echo "legs rc=$?" >> $O/rc.log
OCT_PHMM_DP_ROWS=0 timeout -k 5 300 python tools/long_read_legs.py ccs256x12 > $O/legs_wide.json 2> $O/legs_wide.err
(cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/k -o s -- python /root/repo/tools/long_read_legs.py ccs256x12 ccs2048x12 > /dev/null 2> /root/repo/$O/k.err)
find $O -name "*kernel_trace.csv" -delete
cat $O/rc.log; tail -3 $O/pytest.log; cut -c1-330 $O/legs.json; cut -c1-200 $O/legs_wide.json; head -8 $O/k/s_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
