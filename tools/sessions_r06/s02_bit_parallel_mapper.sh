# Round 6, session 2: the bit-parallel lane mapper (k_kmer_map_lanes), k_scan_finish as one workgroup per count array, k_classify at 6 / 8 waves per SIMD (A/B builds).
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s02; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mapper or populate or fuzz or shapes" > $O/gpu_tests_subset.log 2>&1; echo "tests rc=$?" | tee -a $O/rc.log; tail -3 $O/gpu_tests_subset.log
timeout 300 python bench.py --no-small-batch --no-cpu-baseline --no-extras > $O/bench_default.json 2> $O/bench_default.err
OCT_PHMM_MAP_STATS=1 timeout 300 python bench.py --no-small-batch --no-cpu-baseline --no-extras --steps 1 --warmup 0 2>&1 >/dev/null | grep mapper_pairs | tail -2 | tee $O/map_stats.txt
bash tools/gpu_kernel_split.sh r06_s02 100kx128 stream-hq stream > /dev/null 2>&1
for V in cw6 cw8; do
  OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_$V.so bash tools/gpu_kernel_split.sh r06_s02_$V 100kx128 stream-hq > /dev/null 2>&1
done
python -c "
import json; b=json.load(open('$O/bench_default.json')); print({k:b[k] for k in ('value','ms_per_step','loglik_per_s')})"
for f in gpurun_out/r06_s02*/split_*.txt; do echo "### $f"; head -16 $f | cut -c1-175; done
