cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r03_s40; mkdir -p $O
timeout -k 5 100 python -m pytest tests/test_gpu_parity.py -x -q -k "device_sized or staged or server or basic" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -2 $O/pytest.log
