# Round 5, GPU session 12: the upload's input facts made by the callers' threads - server A/B in one session (same tree, the worker-side pass forced back on by a switch), GPU tests that touch it
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r05_s12; mkdir -p $O
timeout -k 5 600 python -m pytest tests -m gpu -x -q -k "server or shape or contract or host_mirror or patched" 2>&1 | tail -4 > $O/gpu_tests_subset.log
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
{
for rep in 1 2 3; do for SW in "" "OCT_PHMM_SERVER_CALLER_FACTS=0"; do
  echo "## server [$SW]"; env $SW OCT_BENCH_REPS=4 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 16 64 128 2>&1 | grep "\"server\""
done; done
echo "## server [OCT_PHMM_SERVER_PROFILE=1]"; env OCT_PHMM_SERVER_PROFILE=1 OCT_BENCH_REPS=4 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 64 2>&1 | grep "\"server\"\|profile"
echo "## 300x24"; timeout -k 5 200 ./tools/region_calls_bench 3000 300 24 1 16 64 2>&1 | grep "server"
} > $O/server_facts_ab.log 2>&1
tail -2 $O/gpu_tests_subset.log; cat $O/server_facts_ab.log
