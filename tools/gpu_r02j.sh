#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r02j; mkdir -p $O
for sl in 2 3 4 6 8; do
  for mode in 0 1; do
    OCT_PHMM_SLICES=$sl OCT_PHMM_DEDUP=$mode timeout 300 python bench.py --no-cpu-baseline --no-small-batch --steps 10 --warmup 2 --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slices $sl dedup $mode ms', round(d['ms_per_step'],2), 'shared', d['stats']['n_pairs_shared'], 'loglik/s', round(d['loglik_per_s']/1e6,1))"
  done
done | tee $O/slices.txt
