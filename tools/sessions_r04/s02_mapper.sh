# round 4, GPU session 2: the mapper's reduction-free first attempt + mismatch account (pair_mm) - parity, then the kernel split on the four workloads
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r04_s02; mkdir -p $O
timeout -k 5 200 python -m pytest tests/test_gpu_parity.py -x -q -k "basic or mapper or mismatch or golden or generic_bytes or templates or shared" > $O/pytest_subset.log 2>&1; echo "parity subset rc=$?"; tail -1 $O/pytest_subset.log
bash tools/gpu_kernel_split.sh r04_s02 100kx128 100kx128-hq stream stream-hq 2>&1 | grep -v "k_clock_probe\|rocclr\|k_window\|k_hap_tables\|k_scan_tile_sums"
