cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s15; mkdir -p $O
timeout -k 5 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "chunks or linked or basic or late" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
for C in 0 default; do
  if [ $C = default ]; then unset OCT_PHMM_REC_CHUNK; else export OCT_PHMM_REC_CHUNK=$C; fi
  echo "## REC_CHUNK=$C"
  (cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_$C -o s -- python /root/repo/tools/long_read_legs.py ccs-linked > /root/repo/$O/legs_$C.json 2> /root/repo/$O/legs_$C.err)
  cut -c1-260 $O/legs_$C.json; find $O/prof_$C -name "*kernel_trace.csv" -delete; find $O/prof_$C -name "*kernel_stats.csv" | head -1 | xargs head -7 | cut -c1-150
done
