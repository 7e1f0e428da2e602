# Round 6, session 28: the per-rank load of `bench.py --gpus 2` (the N > 1 default: ONE 50,000-region stream, region i on rank i mod N) on one GPU: 25,000 regions resident. Never run on hardware before
# (stream_shard_1_of_8 = 6,250 regions is what every bench line carries). Also the FORCE_DIST path (RCCL init, barrier, reductions at world_size 1).
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_s28; mkdir -p $O
free -g | head -2
OCT_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout -k 5 1200 python bench.py --workload stream --regions 25000 --steps 5 --warmup 1 --no-cpu-baseline --no-small-batch > $O/bench_stream_25000.json 2> $O/bench_stream_25000.err; echo "rc=$?"
cut -c1-1500 $O/bench_stream_25000.json; tail -3 $O/bench_stream_25000.err
