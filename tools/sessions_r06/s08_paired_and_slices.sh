# Round 6, session 8: window pairing (k_pair_sort + PAIRED segments of k_dp) on / off, and 2 ... 5 slices, same box, interleaved; then kernel splits and parity of the paired build.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s08; mkdir -p $O
P="--no-small-batch --no-cpu-baseline --no-extras"
for rep in 1 2; do
for S in 2 3 4 5 8; do for PR in 0 1; do
  OCT_PHMM_PAIRED=$PR OCT_PHMM_SLICES=$S timeout 300 python bench.py $P > $O/b_${S}_${PR}_$rep.json 2> $O/b_${S}_${PR}_$rep.err
  python - <<PY
import json
try:
    b = json.load(open("$O/b_${S}_${PR}_$rep.json")); print("slices $S paired $PR rep $rep", round(b["ms_per_step"], 3), round(b["value"], 1), round(b["roofline"]["avg_launch_ms"], 3), round(b["roofline"]["score_only_kernel_avg_launch_ms"], 3))
except Exception as e: print("$S $PR failed", e)
PY
done; done; done
for S in 3 4 5; do OCT_PHMM_SLICES=$S timeout 300 python bench.py $P --workload stream > $O/s_$S.json 2>/dev/null; python -c "
import json; b=json.load(open('$O/s_$S.json')); print('stream slices $S', round(b['ms_per_step'],3))"; done
# parity of the paired path against the reference's own populate (bench.py verifies 5 % of the matrix)
OCT_PHMM_SLICES=4 timeout 600 python bench.py --no-small-batch --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_verified.json 2> $O/bench_verified.err
python -c "
import json; b=json.load(open('$O/bench_verified.json')); print({k:b.get(k) for k in ('value','ms_per_step','verified_rows','verified_max_abs_diff')})"
OCT_PHMM_PAIRED=1 bash tools/gpu_kernel_split.sh r06_s08_p1 100kx128 > /dev/null 2>&1
OCT_PHMM_PAIRED=0 bash tools/gpu_kernel_split.sh r06_s08_p0 100kx128 > /dev/null 2>&1
for f in gpurun_out/r06_s08_p*/split_*.txt; do echo "### $f"; head -12 $f | cut -c1-175; done
