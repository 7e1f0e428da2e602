# Round 5, GPU session 4: the server with equal-size gathering and asynchronous lone calls, read-record chunks on mid-size batches, then the whole bench.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r05_s04; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
{
for SW in "" "OCT_PHMM_SERVER_GATHER=0" "OCT_PHMM_SERVER_WORKERS=1" "OCT_PHMM_SERVER_WORKERS=3" "OCT_PHMM_REC_CHUNK=64" "OCT_PHMM_SERVER_PIPELINE=0 OCT_PHMM_SERVER_WORKERS=3" ""; do
  echo "## server [$SW]"; env $SW OCT_BENCH_REPS=3 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 16 64 128 2>&1 | grep "\"server\|differ"
done
for SW in "" "OCT_PHMM_SERVER_GATHER=0" "OCT_PHMM_SERVER_PIPELINE=0 OCT_PHMM_SERVER_WORKERS=3"; do
echo "## 300x24 regions [$SW]"; env $SW timeout -k 5 200 ./tools/region_calls_bench 3000 300 24 1 4 16 64 2>&1 | grep "server"
done
} > $O/server_sweep.log 2>&1
{
for N in 16 64; do for SW in "" "OCT_PHMM_REC_CHUNK=64" "OCT_PHMM_REC_CHUNK=96"; do
  echo "## mid_batch $N regions [$SW]"; env $SW timeout -k 5 120 python tools/mid_batch_trace.py $N 2>&1 | tail -1 | cut -c1-200
done; done
} > $O/mid_batch.log 2>&1
timeout -k 5 600 python bench.py > $O/bench.json 2> $O/bench.err
cat $O/server_sweep.log $O/mid_batch.log; python - <<'PY'
import json
d = json.loads(open("/root/repo/gpurun_out/r05_s04/bench.json").read().strip().split("\n")[-1])
print({k: d[k] for k in ("ms_per_step", "value", "region_call_ms", "region_server_regions_per_s", "small_batch_ms", "region_calls_stream_regions_per_s_64_callers")})
print("stream", {k: round(v, 2) for k, v in d["stream"].items() if isinstance(v, float)})
print("stream_hq ms", d["stream_hq"]["ms"], "hq ms", d["hq"]["ms"], "roofline", d["roofline"].get("avg_launch_ms"), d["roofline"].get("score_only_kernel_avg_launch_ms"))
print("cpu_baseline", json.dumps(d["cpu_baseline"])[:1200])
PY
