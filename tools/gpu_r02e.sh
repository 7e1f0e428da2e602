#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r02e; mkdir -p $O
OCT_PHMM_PENALTIES_REPORT=1 timeout 300 python tools/penalty_bench.py > $O/penalty_bench.json 2> $O/err.log
cat $O/penalty_bench.json; grep -v "^$" $O/err.log | sort | uniq -c | tail -12
