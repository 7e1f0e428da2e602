cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu -k "mapper or config2 or slices or basic or fuzz or random_scenarios or 100k or server_batches or stream or ragged" > gpurun_out/r02m_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02m_pytest.log
bash tools/gpu_ab.sh r02m OCT_PHMM_LIB=/root/repo/octopus_amd/variants/v7.so
