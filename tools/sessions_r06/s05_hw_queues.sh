# Round 6, session 5: eight slice streams share four hardware queues by default (profiles/r06_s03: a slice's front end waits behind the DP chain of the slice four before it).
# GPU_MAX_HW_QUEUES=8 against the default, equal slices and one ramp; and the lane mapper's second look (mapper statistics, kernel split).
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s05; mkdir -p $O
P="--no-small-batch --no-cpu-baseline --no-extras"
for rep in 1 2; do
for Q in 4 8; do for R in 1,1,1,1,1,1,1,1 1,2,4,8,8,5,3,1; do
  GPU_MAX_HW_QUEUES=$Q OCT_PHMM_SLICE_RAMP=$R timeout 300 python bench.py $P > $O/b_${Q}_${R}_$rep.json 2> $O/b_${Q}_${R}_$rep.err
  GPU_MAX_HW_QUEUES=$Q OCT_PHMM_SLICE_RAMP=$R timeout 300 python bench.py $P --workload stream > $O/s_${Q}_${R}_$rep.json 2> $O/s_${Q}_${R}_$rep.err
  python - <<PY
import json
for t in ("b", "s"):
    try:
        b = json.load(open("$O/%s_${Q}_${R}_$rep.json" % t)); print("queues $Q ramp $R rep $rep", t, round(b["ms_per_step"], 3), round(b["value"], 1))
    except Exception as e: print("$Q $R", t, "failed", e)
PY
done; done; done
OCT_PHMM_SLICE_RAMP=1,1,1,1,1,1,1,1 OCT_PHMM_MAP_STATS=1 timeout 300 python bench.py $P --steps 1 --warmup 0 2>&1 >/dev/null | grep mapper_pairs | tail -1
OCT_PHMM_SLICE_RAMP=1,1,1,1,1,1,1,1 bash tools/gpu_kernel_split.sh r06_s05 100kx128 stream-hq > /dev/null 2>&1
for f in gpurun_out/r06_s05/split_*.txt; do echo "### $f"; head -9 $f | cut -c1-175; done
