cd /root/repo
timeout 600 python -m pytest tests -x -q -m gpu -k "mapper or config2 or slices or basic or fuzz or random_scenarios or 100k or server_batches" > gpurun_out/r02c_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02c_pytest.log
bash tools/gpu_ab.sh r02c OCT_PHMM_WAVE_MAPPER=1
