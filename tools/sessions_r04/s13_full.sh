# round 4, GPU session 13: whole GPU suite + full default bench line on the tree about to be committed
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r04_s13; mkdir -p $O
timeout -k 5 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -3 $O/pytest_gpu.log
timeout -k 5 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err; cut -c1-300 $O/bench.json
