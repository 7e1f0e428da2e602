cd /root/repo
timeout 600 python -m pytest tests -x -q -m gpu -k "mapper or config2 or slices or basic or fuzz or random_scenarios or 100k or server_batches or stream" > gpurun_out/r02d_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02d_pytest.log
bash tools/gpu_ab.sh r02d OCT_PHMM_MAP_COUNT_ONLY=1
