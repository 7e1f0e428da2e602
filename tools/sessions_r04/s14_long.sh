cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r04_s14; mkdir -p $O
timeout -k 5 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "linked" > $O/pytest.log 2>&1; echo "linked tests rc=$?"; tail -2 $O/pytest.log
(cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof -o s -- python /root/repo/tools/long_read_legs.py > /root/repo/$O/legs.json 2> /root/repo/$O/legs.err); echo "legs rc=$?"; tail -2 $O/legs.err
cut -c1-330 $O/legs.json
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*kernel_stats.csv" | head -1 | xargs head -14 | cut -c1-160
