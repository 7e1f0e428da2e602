# Round 6, session 19: (a) s_setprio for the launches that head a chain (traceback DP, walkers) beside the score-only launch, (b) k_walk_rows fetches the window after the
# staged one while it walks. Variants: shipped (prio 2 + prefetch), noprio, prio3, base (neither). Interleaved: headline / stream / stream-hq, the long-read legs, a 16-region
# populate (mid_batch_trace: upload / run / wait medians), one-region calls and the region server.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s19; mkdir -p $O
P="--no-small-batch --no-cpu-baseline --no-extras"
lib() { [ $1 = default ] && echo "" || echo "OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_$1.so"; }
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "long or region or walk or small" > $O/gpu_tests_subset.log 2>&1; echo "tests rc=$?"; tail -2 $O/gpu_tests_subset.log
for rep in 1 2; do for V in default base noprio prio3; do
  echo "## $V rep $rep"
  env $(lib $V) timeout 300 python tools/long_read_legs.py ccs256x12 ccs2048x12 2>/dev/null | cut -c1-120
  env $(lib $V) timeout 120 python tools/mid_batch_trace.py 16 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('16 regions', j['ms'])"
  env $(lib $V) timeout 120 python tools/mid_batch_trace.py 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' 4 regions', j['ms'])"
  for W in 100kx128 stream-hq; do
    env $(lib $V) timeout 300 python bench.py $P --workload $W > $O/b_${V}_${W}_$rep.json 2> $O/b_${V}_${W}_$rep.err
    python -c "
import json; b=json.load(open('$O/b_${V}_${W}_$rep.json')); print('$V $W rep $rep', round(b['ms_per_step'],3), round(b['value'],1))"
  done
done; done
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
for rep in 1 2; do for V in default base; do
  echo "## server $V rep $rep"
  PRE=""; [ $V != default ] && PRE="LD_PRELOAD=/root/repo/octopus_amd/variants/liboct_phmm_$V.so"
  env $PRE OCT_BENCH_REPS=3 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 1 16 64 2>&1 | grep "\"server\|handle per" | cut -c1-200
done; done
T=/root/repo/$O/trace_16; rm -rf $T
(cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $T -o s -- python /root/repo/tools/mid_batch_trace.py 16 > /dev/null 2>&1)
python tools/timeline_tail.py $T > $O/timeline_16_regions.txt 2>&1; rm -rf $T; cat $O/timeline_16_regions.txt
# the walkers' crossover again (kWalkRowsMaxPairs = 49,152 pairs is round 3's): k_walk_rows (2) against the staged lockstep walker (1) on 16 / 32 / 64 regions in one batch
for N in 16 32 64; do for WS in 1 2; do
  OCT_PHMM_WALK_STAGE=$WS timeout 120 python tools/mid_batch_trace.py $N 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('walk stage $WS, $N regions', j['ms'])"
done; done
for WS in 1 2; do echo "## server walk stage $WS"; OCT_PHMM_WALK_STAGE=$WS OCT_BENCH_REPS=3 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 64 128 2>&1 | grep "\"server" | cut -c1-200; done
# the upload's table kernel inside the first step's k-mer table launch (k_tables) against two launches
for rep in 1 2 3; do for F in 1 0; do
  OCT_PHMM_FUSE_TABLES=$F timeout 120 python tools/mid_batch_trace.py 16 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse tables $F, 16 regions', j['ms'])"
done; done
