# Round 6, session 9: window pairing with ONE gap-word table for both strands (24 bytes per column: the traceback form keeps three workgroups per CU), on / off, interleaved;
# the GPU suite's pairing / mapper / populate tests; kernel split.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s09; mkdir -p $O
P="--no-small-batch --no-cpu-baseline --no-extras"
for rep in 1 2 3; do for PR in 0 1; do
  OCT_PHMM_PAIRED=$PR timeout 300 python bench.py $P > $O/b_${PR}_$rep.json 2> $O/b_${PR}_$rep.err
  python - <<PY
import json
try:
    b = json.load(open("$O/b_${PR}_$rep.json")); print("paired $PR rep $rep", round(b["ms_per_step"], 3), round(b["value"], 1), round(b["roofline"]["avg_launch_ms"], 3), round(b["roofline"]["score_only_kernel_avg_launch_ms"], 3))
except Exception as e: print("$PR failed", e)
PY
done; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "window_paired or mapper or populate or fuzz or shapes" > $O/gpu_tests_subset.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests_subset.log
timeout 600 python bench.py --no-small-batch --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_verified.json 2> $O/bench_verified.err
python -c "
import json; b=json.load(open('$O/bench_verified.json')); print({k:b.get(k) for k in ('value','ms_per_step','verified_rows','verified_max_abs_diff')})"
bash tools/gpu_kernel_split.sh r06_s09 100kx128 > /dev/null 2>&1
for f in gpurun_out/r06_s09/split_*.txt; do echo "### $f"; head -14 $f | cut -c1-175; done
