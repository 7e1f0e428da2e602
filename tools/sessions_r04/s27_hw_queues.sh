cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s27; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
for Q in "" 8 16; do for W in 3 6; do
  echo "## GPU_MAX_HW_QUEUES=$Q workers=$W" >> $O/q.log
  if [ -n "$Q" ]; then export GPU_MAX_HW_QUEUES=$Q; else unset GPU_MAX_HW_QUEUES; fi
  OCT_PHMM_SERVER_WORKERS=$W timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 64 128 2>&1 | grep "\"server\"" >> $O/q.log
done; done
cat $O/q.log
