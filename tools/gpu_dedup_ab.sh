# Exact de-duplication of pairs on and off, over slice counts, bench batch + stream, with the bench's own verification:
#   bash tools/gpu_dedup_ab.sh <tag>      -> gpurun_out/<tag>/dedup_ab.txt
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/${1:-dedup_ab}; mkdir -p $O
show() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['stats']; st = d.get('stream', {})
print('$1', 'ms', round(d['ms_per_step'], 2), 'computed GCUPS', round(d['value'], 1), 'reference-work GCUPS', round(d['gcups_reference_work'], 1), 'loglik/s', round(d['loglik_per_s'] / 1e6, 1),
      'shared pairs', s.get('n_pairs_shared'), 'verified', d.get('verified_rows'), d.get('verified_max_abs_diff'),
      '| stream ms', st.get('ms'), 'regions/s', st.get('regions_per_s'), 'shared', st.get('pairs_shared'), 'verified', st.get('verified_rows'), st.get('verified_max_abs_diff'), 'e2e ms', d.get('e2e_ms_from_host'))"; }
{
for mode in 0 1; do
  OCT_PHMM_DEDUP=$mode timeout 600 python bench.py --no-cpu-baseline --no-small-batch --steps 10 --warmup 2 2> $O/bench_dedup$mode.err | tee $O/bench_dedup$mode.json | show "dedup=$mode"
done
for sl in 2 4 8; do for mode in 0 1; do
  OCT_PHMM_SLICES=$sl OCT_PHMM_DEDUP=$mode timeout 300 python bench.py --no-cpu-baseline --no-small-batch --steps 10 --warmup 2 --no-extras 2>/dev/null | show "slices=$sl dedup=$mode"
done; done
} | tee $O/dedup_ab.txt
