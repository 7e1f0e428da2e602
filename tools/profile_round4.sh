# One gpurun call's worth of tests + bench + profiles for round 4 (summaries are copied into profiles/ afterwards):
#   bash tools/profile_round4.sh <tag> [notests]
# Every command under `timeout -k`; counters in their own passes (--pmc never together with a trace option).
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/${1:-r04_prof}; mkdir -p $O
python -c "from octopus_amd import engine; print(engine.kernel_source_sha())" > $O/kernel_source_sha
if [ "$2" != "notests" ]; then
  timeout -k 5 900 python -m pytest tests -x -q -m gpu --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
fi
timeout -k 5 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
export OCT_PHMM_SLICES=1
P="--no-small-batch --no-cpu-baseline --no-extras"
# kernel stats (single slice) of the four workloads
bash tools/gpu_kernel_split.sh ${1:-r04_prof} 100kx128 100kx128-hq stream stream-hq > $O/split.log 2>&1
(cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/kstats -o s -- python /root/repo/bench.py $P --steps 3 --warmup 1 > /root/repo/$O/bench_1slice_rocprof.json 2> /root/repo/$O/kstats.err); echo "kstats rc=$?" >> $O/rc.log
find $O -name "*kernel_trace.csv" -size +1M -delete
# the four counter passes of the headline workload (bench.py's roofline block cites them)
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
  D=pmc_$(echo $C | cut -d' ' -f1)
  (cd /tmp && timeout -k 5 200 rocprofv3 --pmc $C --output-format csv -d /root/repo/$O/$D -o p -- python /root/repo/bench.py $P --steps 2 --warmup 1 > /root/repo/$O/$D.json 2> /root/repo/$O/$D.err); echo "$D rc=$?" >> $O/rc.log
done
# the counter summary bench.py's roofline block reads (profiles/r*_pmc_summary.json, stamped with the kernel sources' digest): written here so that the bench run below cites counters of these very kernels
python tools/summarize_pmc.py $O/pmc_SQ_WAVES $O/pmc_SQ_ACTIVE_INST_VALU $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE --out $O/${SUMMARY:-r04_final_pmc_summary} --sha $(cat $O/kernel_source_sha) > $O/summarize.log 2>&1; echo "summarize rc=$?" >> $O/rc.log
cp $O/${SUMMARY:-r04_final_pmc_summary}.json $O/${SUMMARY:-r04_final_pmc_summary}.md profiles/ 2>> $O/summarize.log
# mapper + classifier wait split, before (round-3 kernels: tools/next_round/liboct_phmm_head.so) and after, on the headline batch and on stream-hq
for W in 100kx128 stream-hq; do for V in before after; do
  if [ $V = before ]; then export OCT_PHMM_LIB=/root/repo/tools/next_round/liboct_phmm_head.so; else unset OCT_PHMM_LIB; fi
  if [ $V = before ] && [ ! -f /root/repo/tools/next_round/liboct_phmm_head.so ]; then continue; fi
  D=map_pmc_${W}_$V
  (cd /tmp && timeout -k 5 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /root/repo/$O/$D -o p -- python /root/repo/bench.py $P --workload $W --steps 1 --warmup 1 > /root/repo/$O/$D.json 2> /root/repo/$O/$D.err); echo "$D rc=$?" >> $O/rc.log
done; done
unset OCT_PHMM_LIB OCT_PHMM_SLICES
timeout -k 5 300 python tools/stream_e2e.py 1 2 3 > $O/stream_e2e.json 2> $O/stream_e2e.err; echo "stream_e2e rc=$?" >> $O/rc.log
timeout -k 5 200 python tools/long_read_legs.py > $O/long_legs.json 2> $O/long_legs.err; echo "long legs rc=$?" >> $O/rc.log
# the long-read kernels' counters (k_dp_wide<16> on ccs256x12, k_dp_mw<256> on long512x8): instruction mix / waits, VALU busy cycles
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  D=long_pmc_$(echo $C | cut -d' ' -f1)
  (cd /tmp && timeout -k 5 200 rocprofv3 --pmc $C --output-format csv -d /root/repo/$O/$D -o p -- python /root/repo/tools/long_read_legs.py ccs256x12 long512x8 > /root/repo/$O/$D.json 2> /root/repo/$O/$D.err); echo "$D rc=$?" >> $O/rc.log
done
python tools/summarize_pmc.py $O/long_pmc_SQ_WAVES $O/long_pmc_SQ_ACTIVE_INST_VALU --min-grid 1000 --out $O/r04_final_long_read_pmc --sha $(cat $O/kernel_source_sha) >> $O/summarize.log 2>&1
(cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/long_kstats -o s -- python /root/repo/tools/long_read_legs.py ccs256x12 long512x8 > /root/repo/$O/long_kstats.json 2> /root/repo/$O/long_kstats.err); echo "long kstats rc=$?" >> $O/rc.log
find $O -name "*kernel_trace.csv" -size +1M -delete
timeout -k 5 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
du -sh $O; cat $O/rc.log; tail -4 $O/pytest_gpu.log; cut -c1-300 $O/bench.json
