#!/bin/bash
# round 2, wave-per-haplotype penalty kernel: corpus parity on both device kernels, then the three-way timing
set -x
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r02c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "penalty" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.log
OCT_PHMM_PENALTIES_REPORT=1 timeout 300 python tools/penalty_bench.py > $O/penalty_bench.json 2> $O/penalty_bench.err; echo "penalty rc=$?" >> $O/rc.log
cat $O/rc.log; tail -8 $O/pytest.log; cat $O/penalty_bench.json; tail -5 $O/penalty_bench.err
