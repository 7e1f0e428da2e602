# Round 5, GPU session 1: the GPU suite on the new chain (k_scan_fused + joint traceback launches), a slice of the shape fuzz, and the region server / mid-size batches
# with the round-4 chain (OCT_PHMM_SCAN_FUSED=0) beside the new one.   bash tools/sessions_r05/s01_first_look.sh
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r05_s01; mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_tests.log
timeout -k 5 400 python tools/gpu_fuzz.py shapes 400 0 12 > $O/shapes_400.log 2>&1
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
{
for N in 8 16 64; do
  for SW in "" "OCT_PHMM_SCAN_FUSED=0" "OCT_PHMM_LATE_MIN_PAIRS=0" "OCT_PHMM_DEVICE_SIZED=1" "OCT_PHMM_LANE_MAPPER=1" "OCT_PHMM_DEVICE_SIZED=1 OCT_PHMM_LATE_MIN_PAIRS=0"; do
    echo "## mid_batch $N regions [$SW]"; env $SW timeout -k 5 120 python tools/mid_batch_trace.py $N 2>&1 | tail -1
  done
done
} > $O/mid_batch.log 2>&1
{
for SW in "" "OCT_PHMM_SCAN_FUSED=0" "OCT_PHMM_LATE_MIN_PAIRS=0" "OCT_PHMM_LATE_MIN_PAIRS=1000000000" "OCT_PHMM_SERVER_WORKERS=4" "OCT_PHMM_SERVER_WORKERS=2" "OCT_PHMM_LANE_MAPPER=1"; do
  echo "## server [$SW]"; env $SW OCT_BENCH_REPS=3 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 16 64 128 2>&1 | grep "\"server\|differ"
done
} > $O/server_ab.log 2>&1
# timelines: one 16-region populate, new chain and old
for N in 12 16; do for SW in new OCT_PHMM_SCAN_FUSED=0; do
  T=/root/repo/$O/trace_${N}_$SW; rm -rf $T
  (cd /tmp && env ${SW/new/OCT_X=1} timeout -k 5 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $T -o s -- python /root/repo/tools/mid_batch_trace.py $N > /dev/null 2>&1)
  python tools/timeline_tail.py $T > $O/timeline_${N}_${SW/OCT_PHMM_SCAN_FUSED=0/old}.txt 2>&1
  find $T -name "*.csv" -size +2M -delete
done; done
timeout -k 5 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -3 $O/gpu_tests.log; tail -2 $O/shapes_400.log; cat $O/server_ab.log; grep -h "^##\|populate" $O/mid_batch.log | cut -c1-250 | tail -40; tail -4 $O/timeline_16_new.txt; tail -1 $O/bench.json | cut -c1-600
