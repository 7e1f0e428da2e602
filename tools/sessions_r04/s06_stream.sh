# round 4, GPU session 6: page-locked caller buffers - parity tests, then the stream leg of the bench (from-host rates) via tools/stream_e2e.py
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r04_s06; mkdir -p $O
timeout -k 5 200 python -m pytest tests/test_gpu_parity.py -x -q -k "page_locked or one_shot or streamed or basic" > $O/pytest_subset.log 2>&1; echo "parity subset rc=$?"; tail -2 $O/pytest_subset.log
timeout -k 5 300 python tools/stream_e2e.py 1 2 3 > $O/stream_e2e.json 2> $O/stream_e2e.err; echo "stream_e2e rc=$?"; tail -2 $O/stream_e2e.err; cat $O/stream_e2e.json
