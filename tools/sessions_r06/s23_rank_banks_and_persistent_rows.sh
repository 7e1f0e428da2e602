# Round 6, session 23: k_kmer_map_big's rank tail with an odd number of counter words per thread (no 64-way bank conflict), with and without the ballot votes (variant noballots);
# k_dp_rows persistent (one workgroup per CU, waves stride over the groups) for small long-read batches against the covering launch (OCT_PHMM_ROWS_PERSISTENT=0).
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s23; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "big or 40k or long or narrow" > $O/gpu_tests_subset.log 2>&1; echo "tests rc=$?"; tail -2 $O/gpu_tests_subset.log
for rep in 1 2; do for V in default noballots nopersist; do L=""; [ $V = noballots ] && L="OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_noballots.so"; [ $V = nopersist ] && L="OCT_PHMM_ROWS_PERSISTENT=0"; echo "## $V rep $rep"
  env $L timeout 300 python tools/long_read_legs.py ccs256x12 ccs2048x12 long64x8 2>/dev/null | cut -c1-100
done; done
for V in default noballots; do L=""; [ $V = noballots ] && L="OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_noballots.so"
for L2 in ccs256x12 ccs2048x12; do
  T=/root/repo/$O/trace_$L2; rm -rf $T
  (cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $T -o s -- env $L OCT_TRACE_MARK=1 python /root/repo/tools/long_read_legs.py $L2 > /dev/null 2>&1)
  python tools/timeline_tail.py $T > $O/timeline_${V}_$L2.txt 2>&1; rm -rf $T; echo "## $V $L2"; cat $O/timeline_${V}_$L2.txt | cut -c1-150
done; done
