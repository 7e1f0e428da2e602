cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r03_s38; mkdir -p $O
timeout -k 5 150 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -2 $O/pytest.log
