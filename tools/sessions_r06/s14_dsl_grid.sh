# Round 6, session 14: grid cap of the device-sized DP launches (1,024 workgroups that stride over the task groups, against 4,096 / 16,384 that take one unit each): 12- and 16-region batches, region call, region server.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s14; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
for rep in 1 2; do for V in default dsl4096 dsl16384; do
  L=""; [ $V != default ] && L="OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_$V.so"
  for N in 4 12 16 48; do echo "$V rep $rep N=$N $(env $L timeout -k 5 120 python tools/mid_batch_trace.py $N 2>&1 | tail -1 | cut -c1-190)"; done
done; done
# the region server links liboct_phmm.so by rpath: swap the library file for the variant runs
cp octopus_amd/liboct_phmm.so /tmp/liboct_default.so
for V in default dsl4096 dsl16384; do
  [ $V != default ] && cp octopus_amd/variants/liboct_phmm_$V.so octopus_amd/liboct_phmm.so
  echo "## server $V"; OCT_BENCH_REPS=4 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 16 64 2>&1 | grep "\"server" | cut -c1-170
  cp /tmp/liboct_default.so octopus_amd/liboct_phmm.so
done
T=/tmp/trace_dsl; rm -rf $T
(cd /tmp && OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_dsl4096.so timeout -k 5 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $T -o s -- python /root/repo/tools/mid_batch_trace.py 12 > /dev/null 2>&1)
python tools/timeline_tail.py $T | cut -c1-110; rm -rf $T
