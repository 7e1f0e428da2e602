# round 4, GPU session 4: the lane-per-pair mapper (k_kmer_map_lanes) - parity, then the kernel split on the four workloads, then its A/B switch
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r04_s04; mkdir -p $O
timeout -k 5 250 python -m pytest tests/test_gpu_parity.py -x -q -k "basic or mapper or mismatch or golden or generic_bytes or templates or shared or lane" > $O/pytest_subset.log 2>&1; echo "parity subset rc=$?"; tail -3 $O/pytest_subset.log
bash tools/gpu_kernel_split.sh r04_s04 100kx128 100kx128-hq stream stream-hq 2>&1 | grep -v "k_clock_probe\|rocclr\|k_window\|k_hap_tables\|k_scan_tile_sums"
OCT_PHMM_MAP_STATS=1 OCT_PHMM_ENV_SWITCHES=1 timeout -k 5 100 python bench.py --no-cpu-baseline --no-small-batch --steps 2 --warmup 1 --no-extras 2>&1 | grep mapper_pairs | tail -1
OCT_PHMM_MAP_STATS=1 OCT_PHMM_ENV_SWITCHES=1 timeout -k 5 100 python bench.py --workload stream-hq --no-cpu-baseline --no-small-batch --steps 2 --warmup 1 --no-extras 2>&1 | grep mapper_pairs | tail -1
