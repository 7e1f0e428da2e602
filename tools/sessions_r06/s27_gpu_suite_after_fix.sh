# Round 6, session 27: the task-less-row fix of k_walk_rows' prefetch: the repro, the GPU suite, a fuzz subset with host-sized launches in the switch sets, smoke.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s27; mkdir -p $O
timeout 300 python tools/repro_fuzz77.py gpu modes > $O/repro.log 2>&1; echo "repro rc=$?"; tail -1 $O/repro.log | cut -c1-200
timeout 2700 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
timeout -k 5 900 python tools/gpu_fuzz.py shapes 1000 41000 12 > $O/gpu_fuzz_shapes_1000.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/gpu_fuzz_shapes_1000.log | cut -c1-300
timeout -k 5 600 python tools/gpu_fuzz.py 10 50 42000 > $O/gpu_fuzz_small_500.log 2>&1; echo "small fuzz rc=$?"; tail -2 $O/gpu_fuzz_small_500.log | cut -c1-300
OCT_PHMM_DEVICE_SIZED=0 timeout -k 5 600 python tools/gpu_fuzz.py 10 50 43000 > $O/gpu_fuzz_small_500_host_sized.log 2>&1; echo "small fuzz, host-sized launches rc=$?"; tail -2 $O/gpu_fuzz_small_500_host_sized.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
