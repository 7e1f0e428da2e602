# Round 6, session 1: what would the DP loops cost if both packed tasks of a lane shared ONE haplotype window (gap words pre-packed, probe 2: nuc_prior folded into
# the insertion's own gap words)? TIMING ONLY - the probe libraries compute other numbers than the product. Shipped library first, same box.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s01; mkdir -p $O
P="--no-small-batch --no-cpu-baseline --no-extras"
for V in default paired1 paired2; do
  L=""; [ $V != default ] && L="OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_$V.so"
  env $L timeout 300 python bench.py $P > $O/bench_$V.json 2> $O/bench_$V.err
  (cd /tmp && env $L OCT_PHMM_SLICES=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/k_$V -o s -- python /root/repo/bench.py $P --steps 3 --warmup 1 > /root/repo/$O/bench_1slice_$V.json 2> /root/repo/$O/k_$V.err)
  find $O/k_$V -name "*kernel_trace.csv" -delete
  echo "== $V"; python -c "
import json
b=json.load(open('$O/bench_$V.json')); print({k:b[k] for k in ('value','ms_per_step')}, b['roofline']['avg_launch_ms'], b['roofline']['score_only_kernel_avg_launch_ms'])"
  head -6 $(find $O/k_$V -name "*kernel_stats.csv") | cut -c1-160
done
