# A/B of environment switches on the GPU box: bash tools/gpu_ab.sh <tag> "VAR=1" "VAR2=x" ... (first run = no switch)
set -x
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/$1; mkdir -p $O; shift
P="--no-small-batch --no-cpu-baseline --no-extras"
timeout 300 python bench.py $P > $O/bench_default.json 2> $O/bench_default.err
for V in "$@"; do
  N=$(echo $V | tr -c 'A-Za-z0-9_=\n' '_')
  env $V timeout 300 python bench.py $P > $O/bench_$N.json 2> $O/bench_$N.err
  env $V timeout 300 python bench.py $P --workload stream > $O/bench_stream_$N.json 2> $O/bench_stream_$N.err
done
timeout 300 python bench.py $P --workload stream > $O/bench_stream_default.json 2> $O/bench_stream_default.err
export OCT_PHMM_SLICES=1
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/kstats -o s -- python /root/repo/bench.py $P --steps 3 --warmup 1 > /root/repo/$O/bench_1slice_rocprof.json 2> /root/repo/$O/kstats.err)
find $O -name "*kernel_trace.csv" -delete
unset OCT_PHMM_SLICES
for f in $O/bench*.json; do echo $f; python -c "
import json,sys
b=json.load(open('$f')); print({k:b[k] for k in ('value','ms_per_step','loglik_per_s')}, b['roofline']['avg_launch_ms'], b['roofline']['score_only_kernel_avg_launch_ms'])"; done
head -8 $O/kstats/s_kernel_stats.csv | cut -c1-150
