# Round 6, session 18: 6,000 more multi-region shape scenarios (seeds 20000...) and 1,000 small ones on the final tree.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s18; mkdir -p $O
timeout -k 5 2400 python tools/gpu_fuzz.py shapes 6000 20000 12 > $O/gpu_fuzz_shapes_6000.log 2>&1; echo "fuzz shapes rc=$?"
timeout -k 5 900 python tools/gpu_fuzz.py 20 50 6000 > $O/gpu_fuzz_small_1000.log 2>&1; echo "fuzz small rc=$?"
tail -2 $O/gpu_fuzz_shapes_6000.log | cut -c1-400; tail -1 $O/gpu_fuzz_small_1000.log
