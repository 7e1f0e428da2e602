cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r03_s42; mkdir -p $O
timeout -k 5 195 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?"
tail -2 $O/gpu_tests.log
