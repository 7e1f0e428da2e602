cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s45; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -x -q -k "canonical or shar" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.log
timeout -k 5 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "stream" >> $O/pytest.log 2>&1; echo "fullsize rc=$?" >> $O/rc.log
bash tools/gpu_kernel_split.sh r04_s45 stream > $O/split.log 2>&1
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
ok=0; bad=0
for R in $(seq 1 10); do
  OCT_PHMM_SERVER_WORKERS=1 OCT_BENCH_REPS=2 timeout -k 5 120 ./tools/region_calls_bench --file /tmp/stream_regions.bin 128 > $O/out.log 2> $O/err.log
  if [ $? -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); fi
done
echo "one worker, 128 callers: ok=$ok bad=$bad" >> $O/rc.log
timeout -k 5 200 python tools/sessions_r04/s42_repro_batches.py > /dev/null 2>&1; echo "random subsets rc=$?" >> $O/rc.log
cat $O/rc.log; tail -2 $O/pytest.log; grep "window\|ms_per_step" $O/split_stream.txt
