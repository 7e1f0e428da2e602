cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s25; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -x -q -k "contract or basic or shar or dedup or region or server" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.log
timeout -k 5 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "stream" >> $O/pytest.log 2>&1; echo "pytest fullsize rc=$?" >> $O/rc.log
for N in 16 64 256; do
  OCT_PHMM_UPLOAD_PROFILE=1 timeout -k 5 100 python tools/mid_batch_trace.py $N 2>&1 | tail -2 | cut -c1-330 >> $O/up.log
done
OCT_PHMM_DEDUP=0 timeout -k 5 100 python tools/mid_batch_trace.py 64 2>&1 | tail -1 | cut -c1-330 >> $O/up.log
bash tools/gpu_kernel_split.sh r04_s25 stream > $O/split.log 2>&1
cat $O/rc.log; tail -2 $O/pytest.log; cat $O/up.log; grep "dedup\|window\|ms_per_step" $O/split_stream.txt
