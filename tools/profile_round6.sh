# One gpurun call's worth of tests + shape fuzz + bench + profiles for round 6 (summaries are copied into profiles/ afterwards):
#   bash tools/profile_round6.sh <tag> [notests]
# Every command under `timeout -k`; counters in their own passes (--pmc never together with a trace option).
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/${1:-r06_final}; mkdir -p $O
python -c "from octopus_amd import engine; print(engine.kernel_source_sha())" > $O/kernel_source_sha
if [ "$2" != "notests" ]; then
  timeout -k 5 900 python -m pytest tests -x -q -m gpu --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
  timeout -k 5 900 python tools/gpu_fuzz.py shapes 2000 10000 12 > $O/gpu_fuzz_shapes.log 2>&1; echo "fuzz shapes rc=$?" >> $O/rc.log
  timeout -k 5 600 python tools/gpu_fuzz.py 10 50 4000 > $O/gpu_fuzz_small.log 2>&1; echo "fuzz small rc=$?" >> $O/rc.log
fi
timeout -k 5 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
export OCT_PHMM_SLICES=1
P="--no-small-batch --no-cpu-baseline --no-extras"
# kernel stats (single slice) of the four workloads
bash tools/gpu_kernel_split.sh ${1:-r06_final} 100kx128 100kx128-hq stream stream-hq > $O/split.log 2>&1
(cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/kstats -o s -- python /root/repo/bench.py $P --steps 3 --warmup 1 > /root/repo/$O/bench_1slice_rocprof.json 2> /root/repo/$O/kstats.err); echo "kstats rc=$?" >> $O/rc.log
find $O -name "*kernel_trace.csv" -size +1M -delete
# the four counter passes of the headline workload (bench.py's roofline block cites them)
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
  D=pmc_$(echo $C | cut -d' ' -f1)
  (cd /tmp && timeout -k 5 200 rocprofv3 --pmc $C --output-format csv -d /root/repo/$O/$D -o p -- python /root/repo/bench.py $P --steps 2 --warmup 1 > /root/repo/$O/$D.json 2> /root/repo/$O/$D.err); echo "$D rc=$?" >> $O/rc.log
done
python tools/summarize_pmc.py $O/pmc_SQ_WAVES $O/pmc_SQ_ACTIVE_INST_VALU $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE --out $O/r06_final_pmc_summary --sha $(cat $O/kernel_source_sha) > $O/summarize.log 2>&1; echo "summarize rc=$?" >> $O/rc.log
cp $O/r06_final_pmc_summary.json $O/r06_final_pmc_summary.md profiles/ 2>> $O/summarize.log
find $O -name "*counter_collection.csv" -size +1M -delete
unset OCT_PHMM_SLICES
# the region server on the stream's regions: rates, where the workers' time goes, and the kernel timeline of a 12- and a 16-region batch (launches per step)
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
{
for SW in "" "OCT_PHMM_SERVER_PROFILE=1"; do
  echo "## server [$SW]"; env $SW OCT_BENCH_REPS=4 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 16 32 64 128 2>&1 | grep "\"server\|differ\|profile"
done
} > $O/server.log 2>&1
for N in 12 16; do
  timeout -k 5 120 python tools/mid_batch_trace.py $N 2>&1 | tail -1 | cut -c1-400 >> $O/mid_batch.log
  T=/root/repo/$O/trace_$N; rm -rf $T
  (cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $T -o s -- python /root/repo/tools/mid_batch_trace.py $N > /dev/null 2>&1)
  python tools/timeline_tail.py $T > $O/timeline_${N}_regions.txt 2>&1; rm -rf $T
done
timeout -k 5 300 python tools/stream_e2e.py 1 2 3 > $O/stream_e2e.json 2> $O/stream_e2e.err; echo "stream_e2e rc=$?" >> $O/rc.log
timeout -k 5 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
du -sh $O; cat $O/rc.log; tail -4 $O/pytest_gpu.log; tail -1 $O/gpu_fuzz_shapes.log | cut -c1-600; tail -1 $O/gpu_fuzz_small.log; cat $O/server.log | grep -v "server vs"; cut -c1-300 $O/bench.json | tail -1
