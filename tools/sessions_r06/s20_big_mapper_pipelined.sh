# Round 6, session 20: (a) k_kmer_map_big walks four k-mers at a time, stage by stage (hash -> bin bounds -> first four entries), against one k-mer after the other (variant bigmap1);
# (b) reads per workgroup of the wave-per-pair mapper (k_kmer_map<3>) on region-server-sized batches; (c) the kernel timeline of a ccs256x12 step.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s20; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "big or 40k or long" > $O/gpu_tests_subset.log 2>&1; echo "tests rc=$?"; tail -2 $O/gpu_tests_subset.log
for rep in 1 2; do for V in default bigmap1; do L=""; [ $V != default ] && L="OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_$V.so"; echo "## $V rep $rep"
  env $L timeout 300 python tools/long_read_legs.py ccs256x12 ccs2048x12 long64x8 long512x8 2>/dev/null | cut -c1-100; done; done
for N in 4 16 64; do for RPB in 16 32 64 128; do
  OCT_PHMM_MAP_READS_PER_BLOCK=$RPB timeout 120 python tools/mid_batch_trace.py $N 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reads per block $RPB, $N regions', j['ms'])"
done; done
for L in ccs256x12 ccs2048x12; do
  T=/root/repo/$O/trace_$L; rm -rf $T
  (cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $T -o s -- env OCT_TRACE_MARK=1 python /root/repo/tools/long_read_legs.py $L > /dev/null 2>&1)
  python tools/timeline_tail.py $T > $O/timeline_$L.txt 2>&1; rm -rf $T; echo "## $L"; cat $O/timeline_$L.txt | cut -c1-150
done
