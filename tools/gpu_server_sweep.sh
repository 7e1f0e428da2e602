# The region server on the configs[3] stream's regions: callers x workers per device (no profiling switches)   bash tools/gpu_server_sweep.sh <tag>
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/${1:-srvsweep}; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
for W in ${WORKERS:-2 4 8}; do
  echo "## workers=$W"
  OCT_PHMM_SERVER_WORKERS=$W timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin ${CALLERS:-1 16 64 128} 2>&1 | grep "\"server\"\|handle per"
done | tee $O/server_sweep.log
