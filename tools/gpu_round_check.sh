# One gpurun call: parity of the tracked k-mer mapper first, then the A/B benches, kernel stats, and the full GPU suite with what is left.
set -x
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/s16; mkdir -p $O
date +%s > $O/t0
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
timeout 240 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mapper or config2 or slices or smoke or populate_basic" > $O/pytest_mapper.log 2>&1; echo "pytest_mapper rc=$?" >> $O/rc.log
timeout 200 python bench.py > $O/bench_track.json 2> $O/bench_track.err; echo "bench_track rc=$?" >> $O/rc.log
OCT_PHMM_KMER_MAP_SWEEP=1 timeout 120 python bench.py --no-cpu-baseline --no-small-batch > $O/bench_sweep.json 2> $O/bench_sweep.err; echo "bench_sweep rc=$?" >> $O/rc.log
(cd /tmp && OCT_PHMM_SLICES=1 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/kstats -o s -- python /root/repo/bench.py --no-small-batch --no-cpu-baseline --steps 3 --warmup 1 > /root/repo/$O/bench_1slice_rocprof.json 2> /root/repo/$O/kstats.err); echo "kstats rc=$?" >> $O/rc.log
find $O -name "*kernel_trace.csv" -delete
OCT_PHMM_MAP_READS_PER_BLOCK=256 timeout 120 python bench.py --no-cpu-baseline --no-small-batch > $O/bench_rpb256.json 2> $O/bench_rpb256.err; echo "bench_rpb256 rc=$?" >> $O/rc.log
timeout 150 python bench.py --workload stream --no-cpu-baseline > $O/bench_stream.json 2> $O/bench_stream.err; echo "bench_stream rc=$?" >> $O/rc.log
timeout 420 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
date +%s > $O/t1
cat $O/rc.log; tail -3 $O/pytest_mapper.log; tail -12 $O/pytest_gpu.log; cut -c1-260 $O/bench_track.json $O/bench_sweep.json $O/bench_rpb256.json $O/bench_stream.json
