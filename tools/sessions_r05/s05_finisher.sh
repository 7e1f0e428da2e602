# Round 5, GPU session 5: gatherer + finisher threads per worker, dense read-record chunks for mid-size batches.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r05_s05; mkdir -p $O
timeout -k 5 600 python -m pytest tests -m gpu -x -q -k "server or shape or host_mirror or patched or error_model" 2>&1 | tail -4 > $O/gpu_tests_subset.log
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
{
for SW in "" "OCT_PHMM_SERVER_WORKERS=3" "OCT_PHMM_SERVER_GATHER=0" "OCT_PHMM_SERVER_WORKERS=1" "OCT_PHMM_SERVER_PIPELINE=0 OCT_PHMM_SERVER_WORKERS=3" "OCT_PHMM_SERVER_PROFILE=1" ""; do
  echo "## server [$SW]"; env $SW OCT_BENCH_REPS=4 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 16 64 128 2>&1 | grep "\"server\|differ\|profile"
done
for SW in "" "OCT_PHMM_SERVER_WORKERS=3"; do
echo "## 300x24 regions [$SW]"; env $SW timeout -k 5 200 ./tools/region_calls_bench 3000 300 24 1 4 16 64 2>&1 | grep "server\|handle per"
done
} > $O/server_sweep.log 2>&1
{
for N in 12 16; do for SW in ""; do
  echo "## mid_batch $N regions [$SW]"; env $SW timeout -k 5 120 python tools/mid_batch_trace.py $N 2>&1 | tail -1 | cut -c1-200
done; done
} > $O/mid_batch.log 2>&1
for N in 12 16; do
  T=/root/repo/$O/trace_$N; rm -rf $T
  (cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $T -o s -- python /root/repo/tools/mid_batch_trace.py $N > /dev/null 2>&1)
  python tools/timeline_tail.py $T > $O/timeline_$N.txt 2>&1; rm -rf $T
done
tail -2 $O/gpu_tests_subset.log; cat $O/server_sweep.log $O/mid_batch.log $O/timeline_16.txt; tail -2 $O/timeline_12.txt
