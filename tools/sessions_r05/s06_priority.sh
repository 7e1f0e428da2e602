# Round 5, GPU session 6: flavour hints from the callers' threads, the handle's main stream at high priority (server, streaming from host).
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r05_s06; mkdir -p $O
timeout -k 5 600 python -m pytest tests -m gpu -x -q -k "server or shape or host_mirror or stream" 2>&1 | tail -4 > $O/gpu_tests_subset.log
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
{
for SW in "" "OCT_PHMM_STREAM_PRIORITY=0" "" "OCT_PHMM_STREAM_PRIORITY=0"; do
  echo "## server [$SW]"; env $SW OCT_BENCH_REPS=4 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 16 64 128 2>&1 | grep "\"server\|differ\|profile"
done
for SW in "" "OCT_PHMM_STREAM_PRIORITY=0"; do
echo "## 300x24 regions [$SW]"; env $SW timeout -k 5 200 ./tools/region_calls_bench 3000 300 24 1 16 64 2>&1 | grep "server\|handle per"
done
} > $O/server_sweep.log 2>&1
for SW in OCT_X=1 OCT_PHMM_STREAM_PRIORITY=0; do echo "## stream_e2e [$SW]"; env $SW timeout -k 5 300 python tools/stream_e2e.py 1 2 3 2>/dev/null | tail -1 | cut -c1-1500; done > $O/stream_e2e.log 2>&1
for N in 16; do for SW in OCT_X=1 OCT_PHMM_STREAM_PRIORITY=0; do
  T=/root/repo/$O/trace_$N_$SW; rm -rf $T
  (cd /tmp && env $SW timeout -k 5 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $T -o s -- python /root/repo/tools/mid_batch_trace.py $N > /dev/null 2>&1)
  python tools/timeline_tail.py $T > $O/timeline_${N}_$SW.txt 2>&1; rm -rf $T
done; done
tail -2 $O/gpu_tests_subset.log; cat $O/server_sweep.log $O/stream_e2e.log; tail -12 $O/timeline_16_OCT_X=1.txt; tail -10 $O/timeline_16_OCT_PHMM_STREAM_PRIORITY=0.txt
