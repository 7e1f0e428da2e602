# same-box A/B of the committed kernels against tools/next_round/liboct_phmm_lean.so (the saved patch, built on the side; NOT the shipped library)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r03_s41; mkdir -p $O
LEAN=/root/repo/tools/next_round/liboct_phmm_lean.so
OCT_PHMM_LIB=$LEAN timeout -k 5 70 python -m pytest tests/test_gpu_parity.py -x -q -k "basic or staged or late_traceback or device_sized" > $O/pytest_lean.log 2>&1; echo "lean parity rc=$?"; tail -1 $O/pytest_lean.log
for V in head lean; do
  if [ $V = lean ]; then export OCT_PHMM_LIB=$LEAN; else unset OCT_PHMM_LIB; fi
  (cd /tmp && timeout -k 5 45 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace_$V -o s -- python /root/repo/tools/multi_region_trace.py 1 > /root/repo/$O/trace_$V.log 2>&1)
  echo "## $V"; grep "k_walk_rows" $O/trace_$V/s_kernel_stats.csv | cut -d, -f1-4
done
