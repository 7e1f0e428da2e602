cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r04_s37; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
timeout -k 5 400 ./tools/region_calls_bench --file /tmp/stream_regions.bin --out /tmp/out.bin 1 16 32 64 128 > $O/out.log 2> $O/err.log; echo "rc=$?" >> $O/out.log
ls -la /tmp/out.bin >> $O/out.log 2>&1
tail -12 $O/out.log | cut -c1-250; tail -5 $O/err.log
