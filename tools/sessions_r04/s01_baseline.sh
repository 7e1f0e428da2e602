# round 4, GPU session 1: where the step goes on the four workloads (current kernels + the saved k_walk_rows patch applied), and the same-box A/B of that patch
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r04_s01; mkdir -p $O
timeout -k 5 100 python -m pytest tests/test_gpu_parity.py -x -q -k "basic or staged or late_traceback or device_sized or golden" > $O/pytest_subset.log 2>&1; echo "parity subset rc=$?"; tail -1 $O/pytest_subset.log
HEADLIB=/root/repo/tools/next_round/liboct_phmm_head.so
for V in head lean; do
  if [ $V = head ]; then export OCT_PHMM_LIB=$HEADLIB; else unset OCT_PHMM_LIB; fi
  (cd /tmp && timeout -k 5 45 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace_$V -o s -- python /root/repo/tools/multi_region_trace.py 1 > /root/repo/$O/trace_$V.log 2>&1)
  echo "## $V"; grep "k_walk_rows" $O/trace_$V/s_kernel_stats.csv | cut -d, -f1-4; find $O/trace_$V -name "*kernel_trace.csv" -delete
done
unset OCT_PHMM_LIB
bash tools/gpu_kernel_split.sh r04_s01 100kx128 100kx128-hq stream stream-hq
