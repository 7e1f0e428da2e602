# round 4, GPU session 3: the wave mapper with its round-3 first attempt (ladders) + the mismatch account only
cd /root/repo; export TMPDIR=/tmp
bash tools/gpu_kernel_split.sh r04_s03 100kx128 stream-hq 2>&1 | grep "##\|k_kmer_map\|k_classify"
