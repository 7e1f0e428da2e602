cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s30; mkdir -p $O
for N in 6 8 10 12; do for V in "" "OCT_PHMM_DEVICE_SIZED=0"; do
  echo "N=$N [$V] $(env $V timeout -k 5 100 python tools/mid_batch_trace.py $N 2>&1 | tail -1 | cut -c1-260)" >> $O/sweep.log
done; done
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
for V in "" "OCT_PHMM_DEVICE_SIZED=0"; do for R in 1 2; do
  echo "## [$V]" >> $O/sweep.log
  env $V timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 16 64 128 2>&1 | grep "\"server\"" >> $O/sweep.log
done; done
cat $O/sweep.log
