# Round 5, GPU session 7: device-sized DP grids of more, shorter-lived workgroups under the stream priorities (OCT_PHMM_DSL_MAX_BLOCKS)
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r05_s07; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
{
for SW in "" "OCT_PHMM_DSL_MAX_BLOCKS=4096" "OCT_PHMM_DSL_MAX_BLOCKS=16384" "" "OCT_PHMM_DSL_MAX_BLOCKS=4096" "OCT_PHMM_DSL_MAX_BLOCKS=16384"; do
  echo "## server [$SW]"; env $SW OCT_BENCH_REPS=4 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 16 64 128 2>&1 | grep "\"server\|differ"
done
for N in 8 16; do for SW in "" "OCT_PHMM_DSL_MAX_BLOCKS=4096" "OCT_PHMM_DSL_MAX_BLOCKS=16384"; do
  echo "## mid_batch $N regions [$SW]"; env $SW timeout -k 5 120 python tools/mid_batch_trace.py $N 2>&1 | tail -1 | cut -c1-200
done; done
} > $O/blocks.log 2>&1
cat $O/blocks.log
