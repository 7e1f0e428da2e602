# Round 6, session 22: k_kmer_map_big counts the diagonals a wave agrees on with ballots (one LDS atomic per diagonal and wave-step); read hashes in 1,024-base segments and
# 1,024-thread haplotype tables for long inputs; first_mismatch 32 bytes per step; k_dp_rows with unused LDS as a cap on workgroups per CU (OCT_PHMM_ROWS_LDS_KB).
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s22; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "big or 40k or long or region or small" > $O/gpu_tests_subset.log 2>&1; echo "tests rc=$?"; tail -2 $O/gpu_tests_subset.log
for rep in 1 2; do for KB in 0 26 32 40; do echo "## rows lds $KB KB rep $rep"
  OCT_PHMM_ROWS_LDS_KB=$KB timeout 300 python tools/long_read_legs.py ccs256x12 ccs2048x12 2>/dev/null | cut -c1-100
done; done
timeout 300 python tools/long_read_legs.py long64x8 long512x8 ccs-linked 2>/dev/null | cut -c1-100
for N in 1 4 16; do timeout 120 python tools/mid_batch_trace.py $N 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$N regions', j['ms'])"; done
for L in ccs256x12 ccs2048x12; do
  T=/root/repo/$O/trace_$L; rm -rf $T
  (cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $T -o s -- env OCT_TRACE_MARK=1 python /root/repo/tools/long_read_legs.py $L > /dev/null 2>&1)
  python tools/timeline_tail.py $T > $O/timeline_$L.txt 2>&1; rm -rf $T; echo "## $L"; cat $O/timeline_$L.txt | cut -c1-150
done
T=/root/repo/$O/trace_16; rm -rf $T
(cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $T -o s -- python /root/repo/tools/mid_batch_trace.py 16 > /dev/null 2>&1)
python tools/timeline_tail.py $T > $O/timeline_16_regions.txt 2>&1; rm -rf $T; cat $O/timeline_16_regions.txt
P="--no-small-batch --no-cpu-baseline --no-extras"
for W in 100kx128 stream stream-hq; do timeout 300 python bench.py $P --workload $W > $O/b_$W.json 2> $O/b_$W.err; python -c "
import json; b=json.load(open('$O/b_$W.json')); print('$W', round(b['ms_per_step'],3), round(b['value'],1))"; done
