cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r02p; mkdir -p $O
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/small -o s -- python /root/repo/tools/small_trace.py > /root/repo/$O/small.out 2>&1)
cat $O/small/s_kernel_stats.csv | cut -c1-140 | head -24
