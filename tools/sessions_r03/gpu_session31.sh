cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r03_s31; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?"
tail -3 $O/gpu_tests.log
OCT_PHMM_ENV_SWITCHES=1 OCT_PHMM_SERVER_PROFILE=1 timeout 100 ./tools/region_calls_bench 3000 300 24 16 > $O/server_profile.log 2>&1; tail -3 $O/server_profile.log | cut -c1-600
