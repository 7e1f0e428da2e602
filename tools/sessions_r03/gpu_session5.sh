cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "multi_wave or wide or long_reads" > $O/pytest_mw.log 2>&1; echo "pytest rc=$?" >> $O/rc.log
timeout 120 python tools/long_read_run.py 3 > $O/long_read_mw.json 2>$O/long_read_mw.err; echo "long rc=$?" >> $O/rc.log
OCT_PHMM_MULTI_WAVE=0 timeout 120 python tools/long_read_run.py 3 > $O/long_read_1wave.json 2>$O/long_read_1wave.err
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/long_kstats -o s -- python /root/repo/tools/long_read_run.py 2 > /root/repo/$O/long_kstats.json 2>/root/repo/$O/long_kstats.err); echo "long kstats rc=$?" >> $O/rc.log
find $O -name "*kernel_trace.csv" -size +3M -delete
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "long" > $O/pytest_fullsize_long.log 2>&1; echo "fullsize long rc=$?" >> $O/rc.log
cat $O/rc.log; tail -5 $O/pytest_mw.log; cut -c1-400 $O/long_read_mw.json $O/long_read_1wave.json; head -8 $O/long_kstats/s_kernel_stats.csv | cut -c1-160; tail -4 $O/pytest_fullsize_long.log
