# Round 6, session 7: how many slices? 4 / 6 / 8 with the shipped library, 10 / 12 / 16 with a build whose limit is 16 (12.8 M-pair step and the stream).
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s07; mkdir -p $O
P="--no-small-batch --no-cpu-baseline --no-extras"
for rep in 1 2; do
for S in 4 6 8 10 12 16; do
  L=""; [ $S -gt 8 ] && L="OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_s16.so"
  env $L OCT_PHMM_SLICES=$S timeout 300 python bench.py $P > $O/b_${S}_$rep.json 2> $O/b_${S}_$rep.err
  env $L OCT_PHMM_SLICES=$S timeout 300 python bench.py $P --workload stream > $O/s_${S}_$rep.json 2> $O/s_${S}_$rep.err
  python - <<PY
import json
for t in ("b", "s"):
    try:
        b = json.load(open("$O/%s_${S}_$rep.json" % t)); print("slices $S rep $rep", t, round(b["ms_per_step"], 3), round(b["value"], 1))
    except Exception as e: print("$S", t, "failed", e)
PY
done; done
