cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s46; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
OCT_DEBUG_CANON=1 OCT_PHMM_SERVER_WORKERS=1 OCT_BENCH_REPS=1 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 128 > $O/out.log 2> $O/err.log; echo "rc=$?" >> $O/out.log
grep -a "canon check" $O/err.log | awk '{print $3}' | sort | uniq -c | sort -rn | head -5 > $O/summary.txt
grep -a "  region " $O/err.log | head -30 >> $O/summary.txt
grep -a "canon check" $O/err.log | grep -v " 0 bad" | head -6 >> $O/summary.txt
grep -a "bad by" $O/err.log | grep -v ":$" | head -4 | cut -c1-300 >> $O/summary.txt
tail -2 $O/err.log | cut -c1-200 >> $O/summary.txt
cat $O/summary.txt
