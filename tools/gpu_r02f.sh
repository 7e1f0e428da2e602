cd /root/repo
timeout 600 python -m pytest tests -x -q -m gpu -k "mapper or config2 or slices or basic or fuzz or random_scenarios or 100k or server_batches or stream or ragged" > gpurun_out/r02f_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02f_pytest.log
bash tools/gpu_ab.sh r02f OCT_PHMM_LIB=/root/repo/octopus_amd/variants/perm_tables.so
export OCT_PHMM_SLICES=1 OCT_PHMM_LIB=/root/repo/octopus_amd/variants/perm_tables.so
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r02f/kstats_variant -o s -- python /root/repo/bench.py --no-small-batch --no-cpu-baseline --no-extras --steps 3 --warmup 1 > /dev/null 2>&1)
find gpurun_out/r02f -name "*kernel_trace.csv" -delete
head -5 gpurun_out/r02f/kstats_variant/s_kernel_stats.csv | cut -c1-150
