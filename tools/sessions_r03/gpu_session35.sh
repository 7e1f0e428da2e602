cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r03_s35; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python tools/gpu_fuzz.py 30 50 5000 > $O/gpu_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/gpu_fuzz.log
OCT_PHMM_ENV_SWITCHES=1 OCT_PHMM_DEDUP=1 OCT_PHMM_DEDUP_HASH_BITS=3 OCT_PHMM_SLICES=4 timeout 600 python tools/gpu_fuzz.py 10 50 8000 > $O/gpu_fuzz_dedup.log 2>&1; echo "fuzz dedup rc=$?"; tail -2 $O/gpu_fuzz_dedup.log
OCT_PHMM_ENV_SWITCHES=1 OCT_PHMM_WALK_STAGE=2 OCT_PHMM_DSL_MERGE_DP=1 timeout 600 python tools/gpu_fuzz.py 10 50 9000 > $O/gpu_fuzz_rows_merged.log 2>&1; echo "fuzz rows rc=$?"; tail -2 $O/gpu_fuzz_rows_merged.log
cut -c1-300 $O/bench.json
