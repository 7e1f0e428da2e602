# Round 5, GPU session 10: k_scan_finish in rounds (kernel source changed): GPU suite + a fuzz slice, mapper block size A/B on mid-size batches, then the profile round again
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r05_s10; mkdir -p $O
timeout -k 5 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout -k 5 300 python tools/gpu_fuzz.py shapes 500 20000 12 > $O/gpu_fuzz_shapes_500.log 2>&1; echo "fuzz rc=$?" >> $O/rc.log
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
{
for SW in "" "OCT_PHMM_MAP_READS_PER_BLOCK=32" "OCT_PHMM_MAP_READS_PER_BLOCK=64" "OCT_PHMM_MAP_READS_PER_BLOCK=8"; do
  for N in 8 16; do echo "## mid_batch $N regions [$SW]"; env $SW timeout -k 5 120 python tools/mid_batch_trace.py $N 2>&1 | tail -1 | cut -c1-200; done
  echo "## server [$SW]"; env $SW OCT_BENCH_REPS=4 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 16 64 2>&1 | grep "\"server\""
done
} > $O/mapper_block_ab.log 2>&1
bash tools/profile_round5.sh r05_s10 notests > $O/profile_round.log 2>&1
cat $O/rc.log; tail -3 $O/pytest_gpu.log; tail -1 $O/gpu_fuzz_shapes_500.log | cut -c1-300; cat $O/mapper_block_ab.log; grep -h "k_scan_finish\|^##" $O/split_stream-hq.txt $O/split_stream.txt $O/split_100kx128.txt | cut -c1-200; cut -c1-300 $O/bench.json | tail -1
