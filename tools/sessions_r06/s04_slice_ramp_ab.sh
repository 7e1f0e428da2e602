# Round 6, session 4: slice sizes that ramp up and down against eight equal slices (same box, interleaved), 12.8 M-pair step and the 2,000-region stream.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s04; mkdir -p $O
P="--no-small-batch --no-cpu-baseline --no-extras"
for rep in 1 2; do
for R in 1,1,1,1,1,1,1,1 1,2,4,8,8,5,3,1 1,3,6,6,6,6,3,1 1,2,4,8,8,8,4,1 2,4,6,6,6,4,3,1 1,2,3,5,8,8,4,1; do
  OCT_PHMM_SLICE_RAMP=$R timeout 300 python bench.py $P > $O/b_${R}_$rep.json 2> $O/b_${R}_$rep.err
  OCT_PHMM_SLICE_RAMP=$R timeout 300 python bench.py $P --workload stream > $O/s_${R}_$rep.json 2> $O/s_${R}_$rep.err
  python - <<PY
import json
for t in ("b", "s"):
    try:
        b = json.load(open("$O/%s_${R}_$rep.json" % t)); print("$R rep $rep", t, round(b["ms_per_step"], 3), round(b["value"], 1))
    except Exception as e: print("$R", t, "failed", e)
PY
done; done
