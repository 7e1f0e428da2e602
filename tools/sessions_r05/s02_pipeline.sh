# Round 5, GPU session 2: the scan that starts in k_classify + k_scan_finish, device-sized launches up to 400 k pairs, the pipelined region server.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r05_s02; mkdir -p $O
timeout -k 5 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_tests.log
timeout -k 5 300 python tools/gpu_fuzz.py shapes 300 1000 12 > $O/shapes_300.log 2>&1
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
{
for SW in "" "OCT_PHMM_SERVER_PIPELINE=0" "OCT_PHMM_SERVER_WORKERS=2" "OCT_PHMM_SERVER_WORKERS=4" "OCT_PHMM_DEVICE_SIZED=0" "OCT_PHMM_SERVER_PIPELINE=0 OCT_PHMM_DEVICE_SIZED=0 OCT_PHMM_SCAN_FUSED=0" "OCT_PHMM_LATE_MIN_PAIRS=0" "OCT_PHMM_SERVER_PROFILE=1"; do
  echo "## server [$SW]"; env $SW OCT_BENCH_REPS=3 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 16 64 128 2>&1 | grep "\"server\|differ\|profile"
done
} > $O/server_ab.log 2>&1
{
for N in 8 16 64; do
  for SW in "" "OCT_PHMM_SCAN_FUSED=0" "OCT_PHMM_DEVICE_SIZED=0" "OCT_PHMM_JOIN_LATE=0" "OCT_PHMM_WALK_STAGE=0" "OCT_PHMM_WALK_STAGE=2"; do
    echo "## mid_batch $N regions [$SW]"; env $SW timeout -k 5 120 python tools/mid_batch_trace.py $N 2>&1 | tail -1 | cut -c1-330
  done
done
} > $O/mid_batch.log 2>&1
for N in 16; do for SW in new; do
  T=/root/repo/$O/trace_${N}_$SW; rm -rf $T
  (cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $T -o s -- python /root/repo/tools/mid_batch_trace.py $N > /dev/null 2>&1)
  python tools/timeline_tail.py $T > $O/timeline_${N}_$SW.txt 2>&1
  find $T -name "*.csv" -size +2M -delete
done; done
timeout -k 5 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -3 $O/gpu_tests.log; tail -2 $O/shapes_300.log | cut -c1-400; cat $O/server_ab.log; cat $O/mid_batch.log | cut -c1-260; cat $O/timeline_16_new.txt; python - <<'PY'
import json
d = json.loads(open("/root/repo/gpurun_out/r05_s02/bench.json").read().strip().split("\n")[-1])
print({k: d[k] for k in ("ms_per_step", "value", "region_call_ms", "region_server_regions_per_s", "small_batch_ms", "region_calls_stream_regions_per_s_64_callers")})
print("stream", {k: round(v, 2) for k, v in d["stream"].items() if isinstance(v, float)})
print("stream_hq ms", d["stream_hq"]["ms"], "hq ms", d["hq"]["ms"], "roofline", d["roofline"].get("avg_launch_ms"), d["roofline"].get("score_only_kernel_avg_launch_ms"))
print("region_calls", json.dumps(d["region_calls"])[:1500])
PY
