cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s33; mkdir -p $O
for N in 64 256; do for V in "OCT_PHMM_WINDOW_LDS=1" "OCT_PHMM_WINDOW_LDS=0"; do
  echo "N=$N [$V] $(env $V timeout -k 5 100 python tools/mid_batch_trace.py $N 2>&1 | tail -1 | cut -c1-230)" >> $O/sweep.log
done; done
N=64
(cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/k_$N -o s -- python /root/repo/tools/mid_batch_trace.py $N > /root/repo/$O/k_$N.json 2> /root/repo/$O/k_$N.err)
grep "k_window\|k_dedup" $O/k_$N/s_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150 >> $O/sweep.log
find $O -name "*kernel_trace.csv" -delete
cat $O/sweep.log
