# Round 6, session 13: the lane mapper's counting path cuts its hashes out of the code rows; k_kmer_tables no longer writes hash rows (352 B per read). Three workloads, kernel splits, mapper tests.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s13; mkdir -p $O
P="--no-small-batch --no-cpu-baseline --no-extras"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mapper or populate or window_paired" > $O/gpu_tests_subset.log 2>&1; echo "tests rc=$?"; tail -2 $O/gpu_tests_subset.log
for rep in 1 2; do for W in 100kx128 stream stream-hq; do
  timeout 300 python bench.py $P --workload $W > $O/b_${W}_$rep.json 2> $O/b_${W}_$rep.err
  python -c "
import json; b=json.load(open('$O/b_${W}_$rep.json')); print('$W rep $rep', round(b['ms_per_step'],3), round(b['value'],1))"
done; done
bash tools/gpu_kernel_split.sh r06_s13 100kx128 stream-hq > /dev/null 2>&1
for f in gpurun_out/r06_s13/split_*.txt; do echo "### $f"; head -12 $f | cut -c1-175; done
