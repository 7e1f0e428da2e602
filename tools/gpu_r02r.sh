cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r02r; mkdir -p $O
OCT_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --no-small-batch > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "dist rc=$?"; tail -3 $O/bench_dist1.err; cut -c1-300 $O/bench_dist1.json
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
b=json.load(open('/root/repo/gpurun_out/r02r/bench.json'))
print({k:b[k] for k in ('value','ms_per_step','e2e_ms_from_host','small_batch_ms','verified_rows','verified_max_abs_diff')}, b['roofline']['pmc_summary_matches_these_kernels'], b['roofline']['valu']['frac'], b['roofline']['valu']['issue_weighted_frac'], b['roofline']['traffic_ratio'])
PY
timeout 300 python -m pytest tests -x -q -m gpu -k "streamed or slices or one_shot or templates" 2>&1 | tail -2
