# Round 5, GPU session 8: the 2,000-scenario shape fuzz on the final tree (harness keeps at most six engines and two servers alive; pools trim their siblings), and the
# 12.8 M-pair step / the stream with and without the main stream's priority (three interleaved repetitions)
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r05_s08; mkdir -p $O
timeout -k 5 900 python tools/gpu_fuzz.py shapes 2000 10000 12 > $O/gpu_fuzz_shapes.log 2>&1; echo "fuzz shapes rc=$?" > $O/rc.log
P="--no-small-batch --no-cpu-baseline --no-extras"
{
for rep in 1 2 3; do for V in "OCT_X=1" "OCT_PHMM_STREAM_PRIORITY=0"; do
  echo "## headline [$V]"; env $V timeout -k 5 200 python bench.py $P 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k:round(b[k],3) for k in ('value','ms_per_step')}, 'trace', round(b['roofline']['avg_launch_ms'],3), 'score', round(b['roofline']['score_only_kernel_avg_launch_ms'],3))"
done; done
for rep in 1 2; do for V in "OCT_X=1" "OCT_PHMM_STREAM_PRIORITY=0"; do
  echo "## stream-hq [$V]"; env $V timeout -k 5 200 python bench.py $P --workload stream-hq 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k:round(b[k],3) for k in ('value','ms_per_step')})"
done; done
} > $O/priority_big_batches.log 2>&1
cat $O/rc.log; tail -2 $O/gpu_fuzz_shapes.log | cut -c1-900; cat $O/priority_big_batches.log
