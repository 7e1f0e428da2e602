# round 4, GPU session 5: the whole GPU suite, the kernel split of the stream workloads (upload kernels included), one full default bench line
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r04_s05; mkdir -p $O
timeout -k 5 500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -3 $O/pytest_gpu.log
bash tools/gpu_kernel_split.sh r04_s05 stream 100kx128 2>&1 | grep -v "k_clock_probe\|rocclr\|k_scan_tile_sums"
timeout -k 5 420 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err; cut -c1-400 $O/bench.json
