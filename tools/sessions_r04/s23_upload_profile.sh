cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s23; mkdir -p $O
for N in 16 64 256; do
  OCT_PHMM_UPLOAD_PROFILE=1 timeout -k 5 100 python tools/mid_batch_trace.py $N 2>&1 | tail -4 | cut -c1-400 >> $O/up.log
done
cat $O/up.log
