cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s47; mkdir -p $O
./tools/urem_probe > $O/urem.txt 2>&1
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -x -q -k "canonical or shar or server or region" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.log
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
for W in 1 3; do ok=0; bad=0
for R in $(seq 1 8); do
  OCT_PHMM_SERVER_WORKERS=$W OCT_BENCH_REPS=2 timeout -k 5 120 ./tools/region_calls_bench --file /tmp/stream_regions.bin 128 > $O/out.log 2> $O/err.log
  if [ $? -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); fi
done
echo "workers $W, 128 callers: ok=$ok bad=$bad" >> $O/rc.log; done
mkdir -p gpurun_out/r04_s42; timeout -k 5 200 python tools/sessions_r04/s42_repro_batches.py > /dev/null 2>&1; echo "random subsets rc=$?" >> $O/rc.log
cat $O/rc.log; tail -2 $O/pytest.log; cat $O/urem.txt | cut -c1-200
