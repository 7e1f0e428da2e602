# Round 6, session 29: the bench line must be the LAST line on stdout also when RCCL has written its banner through C stdio (session 28 found it behind the line).
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_s29; mkdir -p $O
OCT_BENCH_FORCE_DIST=1 timeout -k 5 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/torchrun_force_dist.out 2> $O/torchrun_force_dist.err; echo "rc=$?"
echo "lines on stdout: $(wc -l < $O/torchrun_force_dist.out)"; tail -1 $O/torchrun_force_dist.out | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('last line parses:', j['metric'], round(j['value']), j['unit'], j['n_gpus'])"
head -c 300 $O/torchrun_force_dist.out; echo
OCT_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29518 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout -k 5 600 python bench.py --workload stream --regions 2000 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/stream_force_dist.out 2> $O/stream_force_dist.err; echo "rc=$?"
tail -1 $O/stream_force_dist.out | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('last line parses:', j['workload'], round(j['value']), j['scaling'])"
timeout -k 5 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('plain run parses:', round(j['value']))"
