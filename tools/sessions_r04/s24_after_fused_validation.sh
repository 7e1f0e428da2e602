cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s24; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -x -q -k "contract or basic or region or server or page_locked" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.log
for N in 16 64 256; do
  OCT_PHMM_UPLOAD_PROFILE=1 timeout -k 5 100 python tools/mid_batch_trace.py $N 2>&1 | tail -2 | cut -c1-330 >> $O/up.log
done
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
for R in 1 2; do timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 16 64 128 2>&1 | grep "\"server\"" >> $O/server.log; done
timeout -k 5 300 python tools/stream_e2e.py 1 3 > $O/stream_e2e.json 2> $O/stream_e2e.err
cat $O/rc.log; tail -2 $O/pytest.log; cat $O/up.log $O/server.log; cut -c1-700 $O/stream_e2e.json
