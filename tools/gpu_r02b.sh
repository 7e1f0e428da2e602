set -x
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r02b; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout 300 python tools/penalty_bench.py > $O/penalty_bench.json 2> $O/penalty_bench.err; echo "penalty rc=$?" >> $O/rc.log
cat $O/rc.log; tail -25 $O/pytest_gpu.log; cat $O/penalty_bench.json; tail -5 $O/penalty_bench.err
