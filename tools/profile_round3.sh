# One gpurun call's worth of tests + bench + profiles for round 3 (summaries are copied into profiles/ afterwards):
#   bash tools/profile_round3.sh <tag> [notests]
set -x
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/${1:-r03_prof}; mkdir -p $O
python -c "from octopus_amd import engine; print(engine.kernel_source_sha())" > $O/kernel_source_sha
if [ "$2" != "notests" ]; then
  timeout -k 5 1200 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
fi
timeout -k 5 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
timeout -k 5 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
timeout -k 5 300 python bench.py --workload stream --no-cpu-baseline --no-small-batch > $O/bench_stream.json 2> $O/bench_stream.err; echo "bench_stream rc=$?" >> $O/rc.log
timeout -k 5 200 ./tools/region_calls_bench 2000 300 24 1 4 8 16 > $O/region_calls.log 2>&1; echo "region_calls rc=$?" >> $O/rc.log
timeout -k 5 100 python tools/latency_breakdown.py > $O/latency.json 2>&1
(cd /tmp && timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/region_trace -o s -- python /root/repo/tools/multi_region_trace.py 1 > /root/repo/$O/region_trace.log 2>&1); echo "region_trace rc=$?" >> $O/rc.log
timeout -k 5 300 python tools/stream_e2e.py 1 2 > $O/stream_e2e.json 2> $O/stream_e2e.err; echo "stream_e2e rc=$?" >> $O/rc.log
export OCT_PHMM_SLICES=1
P="--no-small-batch --no-cpu-baseline --no-extras"
(cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/kstats -o s -- python /root/repo/bench.py $P --steps 3 --warmup 1 > /root/repo/$O/bench_1slice_rocprof.json 2> /root/repo/$O/kstats.err); echo "kstats rc=$?" >> $O/rc.log
find $O -name "*kernel_trace.csv" -size +3M -delete
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
  D=pmc_$(echo $C | cut -d' ' -f1)
  (cd /tmp && timeout -k 5 200 rocprofv3 --pmc $C --output-format csv -d /root/repo/$O/$D -o p -- python /root/repo/bench.py $P --steps 2 --warmup 1 > /root/repo/$O/$D.json 2> /root/repo/$O/$D.err); echo "$D rc=$?" >> $O/rc.log
done
unset OCT_PHMM_SLICES
# BASELINE configs[4] (long reads): kernel stats + the four counter passes of the new kernels
timeout -k 5 120 python tools/long_read_run.py 3 > $O/long_read.json 2>$O/long_read.err; echo "long rc=$?" >> $O/rc.log
(cd /tmp && timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/long_kstats -o s -- python /root/repo/tools/long_read_run.py 2 > /root/repo/$O/long_kstats.json 2>/root/repo/$O/long_kstats.err); echo "long kstats rc=$?" >> $O/rc.log
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" FETCH_SIZE WRITE_SIZE; do
  D=long_pmc_$(echo $C | cut -d' ' -f1)
  (cd /tmp && timeout -k 5 150 rocprofv3 --pmc $C --output-format csv -d /root/repo/$O/$D -o p -- python /root/repo/tools/long_read_run.py 2 > /root/repo/$O/$D.json 2> /root/repo/$O/$D.err); echo "$D rc=$?" >> $O/rc.log
done
find $O -name "*kernel_trace.csv" -size +3M -delete
timeout -k 5 200 ./tools/valu_ubench > $O/valu_ubench.log 2>&1
du -sh $O; cat $O/rc.log; tail -12 $O/pytest_gpu.log; cut -c1-300 $O/bench.json $O/bench_stream.json; cat $O/region_calls.log | cut -c1-200; cat $O/stream_e2e.json | cut -c1-600
