cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02n_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02n_pytest.log
bash tools/gpu_ab.sh r02n OCT_PHMM_LIB=/root/repo/octopus_amd/variants/v8.so
