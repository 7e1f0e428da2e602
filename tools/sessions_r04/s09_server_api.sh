cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s09; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
(cd /tmp && timeout -k 5 200 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d /root/repo/$O/api -o s -- /root/repo/tools/region_calls_bench --file /tmp/stream_regions.bin 64 > /root/repo/$O/run.log 2>&1)
cat $O/run.log | grep "\"server\""
find $O/api -name "*hip_api_stats.csv" | head -1 | xargs head -25 | cut -c1-150
find $O/api -name "*kernel_stats.csv" | head -1 | xargs head -12 | cut -c1-150
find $O/api -name "*_trace.csv" -size +20M -delete
