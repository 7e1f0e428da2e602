# One gpurun call's worth of profiling for a round: see profiles/ for the summaries this produced.
set -x
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/p33; mkdir -p $O
OCT_PHMM_LATE_MIN_PAIRS=99999999999 python bench.py --workload stream --no-cpu-baseline > $O/stream_nolate.json 2> $O/stream_nolate.err
python bench.py --workload stream --no-cpu-baseline > $O/stream_late.json 2> $O/stream_late.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/st_late -o s -- python /root/repo/bench.py --workload stream --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> /root/repo/$O/st_late.err)
(cd /tmp && OCT_PHMM_LATE_MIN_PAIRS=99999999999 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/st_nolate -o s -- python /root/repo/bench.py --workload stream --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> /root/repo/$O/st_nolate.err)
export OCT_PHMM_SLICES=1
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d /root/repo/$O/pmc_$C -o p -- python /root/repo/bench.py --no-small-batch --no-cpu-baseline --steps 2 --warmup 1 > /root/repo/$O/pmc_$C.json 2> /root/repo/$O/pmc_$C.err)
done
find $O -name "*kernel_trace.csv" -delete
du -sh $O; cat $O/stream_nolate.json | cut -c1-200; cat $O/stream_late.json | cut -c1-200
