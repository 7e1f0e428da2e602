# Where the region server's workers spend their time on the configs[3] stream's regions (one call per region from N caller threads):
#   bash tools/gpu_server_profile.sh <tag> [callers ...]     -> gpurun_out/<tag>/server_profile.log
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/${1:-srvprof}; mkdir -p $O; shift
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
for W in ${WORKERS:-2}; do
for C in "${@:-64}"; do
  echo "## workers=$W callers=$C"
  OCT_PHMM_SERVER_WORKERS=$W OCT_PHMM_SERVER_PROFILE=1 timeout -k 5 120 ./tools/region_calls_bench --file /tmp/stream_regions.bin $C 2>&1 | grep "server_profile\|\"server\""
done; done | tee $O/server_profile.log
