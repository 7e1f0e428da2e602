cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r04_s16; mkdir -p $O
timeout -k 5 250 python -m pytest tests/test_gpu_parity.py -x -q -k "basic or mapper or mismatch or lane or shared or templates" > $O/pytest_subset.log 2>&1; echo "parity subset rc=$?"; tail -2 $O/pytest_subset.log
bash tools/gpu_kernel_split.sh r04_s16 100kx128 stream stream-hq 2>&1 | grep "##\|k_kmer_map\|k_classify\|k_hap_bases\|k_kmer_tables"
OCT_PHMM_MAP_STATS=1 OCT_PHMM_ENV_SWITCHES=1 timeout -k 5 100 python bench.py --no-cpu-baseline --no-small-batch --steps 2 --warmup 1 --no-extras 2>&1 | grep mapper_pairs | tail -1
