cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s11; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
for I in 1 2 3 4 5; do
  (cd /tmp && OCT_PHMM_SERVER_WORKERS=3 timeout -k 5 100 rocprofv3 --hip-trace --stats --output-format csv -d /root/repo/$O/api$I -o s -- /root/repo/tools/region_calls_bench --file /tmp/stream_regions.bin 64 > /root/repo/$O/run$I.log 2>&1)
  grep "\"server\"" $O/run$I.log | cut -c1-140
  find $O/api$I -name "*hip_api_stats.csv" | head -1 | xargs head -6 | cut -c1-120
  find $O/api$I -name "*_trace.csv" -delete
done
