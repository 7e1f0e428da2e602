#!/bin/bash
set -x
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r02d; mkdir -p $O
REGIONS=2000 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o pen -- python tools/penalty_bench.py > $O/penalty_bench.json 2> $O/err.log
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {}' 
cat $O/penalty_bench.json
