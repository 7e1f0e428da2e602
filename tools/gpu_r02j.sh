cd /root/repo
timeout 600 python -m pytest tests -x -q -m gpu -k "mapper or config2 or slices or basic or fuzz or random_scenarios or 100k or server_batches or stream or ragged" > gpurun_out/r02j_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02j_pytest.log
OCT_PHMM_MAP_STATS=1 python bench.py --no-small-batch --no-cpu-baseline --no-extras --steps 1 --warmup 0 2>&1 | grep mapper_pairs | head -2
bash tools/gpu_ab.sh r02j OCT_PHMM_LIB=/root/repo/octopus_amd/variants/v3.so
