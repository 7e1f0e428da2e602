# Round 5, GPU session 17 (80 s of budget left): the three compiled seams (the first now with its fallback for refused regions) and the model-file test on the device
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r05_s17; mkdir -p $O
timeout -k 3 75 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "patched or model_file" --durations=4 > $O/pytest_seams.log 2>&1; echo "rc=$?" >> $O/pytest_seams.log
tail -8 $O/pytest_seams.log
