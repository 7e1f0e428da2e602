#!/bin/bash
# round 2, step 4: all GPU tests on the wave penalty kernel build, then the three-way penalty timing
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r02f; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" > $O/rc.log
OCT_PHMM_PENALTIES_REPORT=1 timeout 300 python tools/penalty_bench.py > $O/penalty_bench.json 2> $O/penalty_bench.err; echo "penalty rc=$?" >> $O/rc.log
cat $O/rc.log; tail -12 $O/pytest_gpu.log; cat $O/penalty_bench.json; sort $O/penalty_bench.err | uniq -c | tail -4
