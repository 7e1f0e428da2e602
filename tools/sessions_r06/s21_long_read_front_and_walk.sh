# Round 6, session 21: k_kmer_map_big with one word per bin and 8-byte entry loads; k_walk_rows' prefetch without a select on the loaded values (variant nopf = no prefetch);
# k_dp_rows one wave per workgroup for launches of about a wave per SIMD (OCT_PHMM_ROWS_BLOCK_WAVES=4: four as before). Long-read legs, their timelines, the region-sized legs.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s21; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "big or 40k or long or region or walk or small" > $O/gpu_tests_subset.log 2>&1; echo "tests rc=$?"; tail -2 $O/gpu_tests_subset.log
for rep in 1 2; do for V in default nopf bw4; do L=""; [ $V = nopf ] && L="OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_nopf.so"; [ $V = bw4 ] && L="OCT_PHMM_ROWS_BLOCK_WAVES=4"; echo "## $V rep $rep"
  env $L timeout 300 python tools/long_read_legs.py ccs256x12 ccs2048x12 long64x8 2>/dev/null | cut -c1-100
  env $L timeout 120 python tools/mid_batch_trace.py 1 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' 1 region ', j['ms'])"
  env $L timeout 120 python tools/mid_batch_trace.py 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' 4 regions', j['ms'])"
done; done
for L in ccs256x12 ccs2048x12; do
  T=/root/repo/$O/trace_$L; rm -rf $T
  (cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $T -o s -- env OCT_TRACE_MARK=1 python /root/repo/tools/long_read_legs.py $L > /dev/null 2>&1)
  python tools/timeline_tail.py $T > $O/timeline_$L.txt 2>&1; rm -rf $T; echo "## $L"; cat $O/timeline_$L.txt | cut -c1-150
done
