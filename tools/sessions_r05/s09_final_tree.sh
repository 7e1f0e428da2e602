# Round 5, GPU session 9: the final tree (all slices of a multi-slice batch off the high-priority stream): GPU suite, bench line, kernel stats of the single-slice run, server check
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r05_s09; mkdir -p $O
python -c "from octopus_amd import engine; print(engine.kernel_source_sha())" > $O/kernel_source_sha
timeout -k 5 900 python -m pytest tests -x -q -m gpu --durations=4 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout -k 5 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
P="--no-small-batch --no-cpu-baseline --no-extras"
{
for rep in 1 2; do for V in "OCT_X=1" "OCT_PHMM_STREAM_PRIORITY=0"; do
  echo "## headline [$V]"; env $V timeout -k 5 200 python bench.py $P 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k:round(b[k],3) for k in ('value','ms_per_step')}, 'trace', round(b['roofline']['avg_launch_ms'],3), 'score', round(b['roofline']['score_only_kernel_avg_launch_ms'],3))"
  echo "## stream-hq [$V]"; env $V timeout -k 5 200 python bench.py $P --workload stream-hq 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k:round(b[k],3) for k in ('value','ms_per_step')})"
done; done
} > $O/priority_big_batches.log 2>&1
(cd /tmp && OCT_PHMM_SLICES=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/kstats -o s -- python /root/repo/bench.py $P --steps 3 --warmup 1 > /root/repo/$O/bench_1slice_rocprof.json 2> /root/repo/$O/kstats.err); echo "kstats rc=$?" >> $O/rc.log
find $O -name "*kernel_trace.csv" -size +1M -delete
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
env OCT_BENCH_REPS=4 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 16 64 128 2>&1 | grep "\"server\|differ" > $O/server.log
timeout -k 5 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; tail -3 $O/pytest_gpu.log; cat $O/priority_big_batches.log $O/server.log; cut -c1-400 $O/bench.json | tail -1
