cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s22; mkdir -p $O; nproc > $O/nproc.txt
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
for W in 3 5; do
  echo "## workers=$W" >> $O/host.log
  OCT_PHMM_SERVER_WORKERS=$W timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 16 64 128 256 2>&1 | grep "\"server\"" >> $O/host.log
done
cat $O/nproc.txt $O/host.log
