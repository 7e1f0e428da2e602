# Lean measurement call: smoke, a parity subset, bench (default), single-slice kernel stats, stream bench.
set -x
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${1:-ab}; mkdir -p $O
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
timeout 240 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${2:-golden or random_windows_fast or config2 or slices or random_scenarios or populate_basic or late}" > $O/pytest_subset.log 2>&1; echo "pytest_subset rc=$?" >> $O/rc.log
timeout 200 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench_default rc=$?" >> $O/rc.log
(cd /tmp && OCT_PHMM_SLICES=1 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/kstats -o s -- python /root/repo/bench.py --no-small-batch --no-cpu-baseline --steps 3 --warmup 1 > /root/repo/$O/bench_1slice_rocprof.json 2> /root/repo/$O/kstats.err); echo "kstats rc=$?" >> $O/rc.log
find $O -name "*kernel_trace.csv" -delete
timeout 150 python bench.py --workload stream --no-cpu-baseline > $O/bench_stream.json 2> $O/bench_stream.err; echo "bench_stream rc=$?" >> $O/rc.log
cat $O/rc.log; tail -3 $O/pytest_subset.log; cut -c1-200 $O/bench_default.json $O/bench_stream.json; head -6 $O/kstats/s_kernel_stats.csv | cut -c1-150
