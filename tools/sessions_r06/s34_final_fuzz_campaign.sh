# Round 6, session 34: the same fuzz campaign, fresh seeds, on the kernels that ship (digest 61562fb060fe4830: walk-event words): 4,000 multi-region scenarios with fresh seeds, 2,000 small scenarios, 2,000 small ones with host-sized launches
# (exactly sized traceback scratch: where session 25's fault lived), the long-read checks.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s34; mkdir -p $O
timeout -k 5 2700 python tools/gpu_fuzz.py shapes 4000 90000 12 > $O/gpu_fuzz_shapes_4000.log 2>&1; echo "shapes rc=$?"; tail -1 $O/gpu_fuzz_shapes_4000.log | cut -c1-400
timeout -k 5 900 python tools/gpu_fuzz.py 40 50 61000 > $O/gpu_fuzz_small_2000.log 2>&1; echo "small rc=$?"; tail -1 $O/gpu_fuzz_small_2000.log | cut -c1-300
OCT_PHMM_DEVICE_SIZED=0 timeout -k 5 900 python tools/gpu_fuzz.py 40 50 71000 > $O/gpu_fuzz_small_2000_host_sized.log 2>&1; echo "small host-sized rc=$?"; tail -1 $O/gpu_fuzz_small_2000_host_sized.log | cut -c1-300
OCT_PHMM_DEVICE_SIZED=0 OCT_PHMM_WALK_STAGE=2 timeout -k 5 900 python tools/gpu_fuzz.py 20 50 81000 > $O/gpu_fuzz_small_1000_host_sized_rows.log 2>&1; echo "small host-sized, row walker rc=$?"; tail -1 $O/gpu_fuzz_small_1000_host_sized_rows.log | cut -c1-300
timeout -k 5 600 python tools/long_read_check.py > $O/long_read_check.log 2>&1; echo "long read check rc=$?"; tail -2 $O/long_read_check.log | cut -c1-300
