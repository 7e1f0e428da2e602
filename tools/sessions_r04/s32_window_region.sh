cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s32; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -x -q -k "canonical or shar or contract or basic or region" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.log
timeout -k 5 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "stream" >> $O/pytest.log 2>&1; echo "pytest fullsize rc=$?" >> $O/rc.log
bash tools/gpu_kernel_split.sh r04_s32 stream > $O/split.log 2>&1
timeout -k 5 300 python tools/stream_e2e.py 1 2 3 > $O/stream_e2e.json 2> $O/stream_e2e.err
timeout -k 5 100 python tools/mid_batch_trace.py 64 2>&1 | tail -1 | cut -c1-260 > $O/mid64.log
cat $O/rc.log; tail -2 $O/pytest.log; grep "window\|hap_tables\|ms_per_step\|fillBuffer\|copyBuffer" $O/split_stream.txt; cut -c1-1500 $O/stream_e2e.json; cat $O/mid64.log
