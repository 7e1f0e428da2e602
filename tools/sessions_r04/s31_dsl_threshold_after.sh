cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s31; mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.log
for N in 4 8 16 64 256; do
  echo "N=$N $(OCT_PHMM_UPLOAD_PROFILE=1 timeout -k 5 100 python tools/mid_batch_trace.py $N 2>&1 | tail -2 | cut -c1-250 | tr '\n' ' ')" >> $O/up.log
done
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
for R in 1 2; do timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 16 64 128 2>&1 | grep "\"server\"" >> $O/server.log; done
timeout -k 5 200 ./tools/region_calls_bench 3000 300 24 1 16 2>&1 | grep "mode" | cut -c1-200 >> $O/server.log
cat $O/rc.log; tail -2 $O/pytest.log; cat $O/up.log $O/server.log
