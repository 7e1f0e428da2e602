# One gpurun call's worth of tests + bench + profiles for a round (summaries are copied into profiles/ afterwards):
#   bash tools/profile_round.sh <tag> [notests]   -> gpurun_out/<tag>/{bench*.json, kstats/, pmc_*/, pytest_gpu.log, kernel_source_sha}
set -x
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/${1:-prof}; mkdir -p $O
python -c "from octopus_amd import engine; print(engine.kernel_source_sha())" > $O/kernel_source_sha
if [ "$2" != "notests" ]; then
  timeout 900 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
fi
timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
timeout 420 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
timeout 200 python bench.py --workload stream --no-cpu-baseline --no-small-batch > $O/bench_stream.json 2> $O/bench_stream.err; echo "bench_stream rc=$?" >> $O/rc.log
export OCT_PHMM_SLICES=1
P="--no-small-batch --no-cpu-baseline --no-extras"
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/kstats -o s -- python /root/repo/bench.py $P --steps 3 --warmup 1 > /root/repo/$O/bench_1slice_rocprof.json 2> /root/repo/$O/kstats.err); echo "kstats rc=$?" >> $O/rc.log
find $O -name "*kernel_trace.csv" -delete
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
  D=pmc_$(echo $C | cut -d' ' -f1)
  (cd /tmp && timeout 200 rocprofv3 --pmc $C --output-format csv -d /root/repo/$O/$D -o p -- python /root/repo/bench.py $P --steps 2 --warmup 1 > /root/repo/$O/$D.json 2> /root/repo/$O/$D.err); echo "$D rc=$?" >> $O/rc.log
done
unset OCT_PHMM_SLICES
du -sh $O; cat $O/rc.log; tail -15 $O/pytest_gpu.log; cut -c1-400 $O/bench.json $O/bench_stream.json
