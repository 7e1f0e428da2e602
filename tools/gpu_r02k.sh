cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r02k; mkdir -p $O
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > /root/repo/$O/sq_counters.txt); wc -l $O/sq_counters.txt
export OCT_PHMM_SLICES=1
P="--no-small-batch --no-cpu-baseline --no-extras --steps 1 --warmup 1"
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES"; do
  D=pmc_$(echo $C | cut -d' ' -f2)
  (cd /tmp && timeout 200 rocprofv3 --pmc $C --output-format csv -d /root/repo/$O/$D -o p -- python /root/repo/bench.py $P > /dev/null 2> /root/repo/$O/$D.err); echo "$D rc=$?"
done
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r02k/pmc_*/p_counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(f)):
        if int(r['Grid_Size'])<1000000: continue
        k=r['Kernel_Name'].split('(')[0][-34:]; agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
    for k,c in agg.items():
        if 'kmer_map' in k or 'k_dp' in k or 'walk' in k or 'classify' in k: print(k, {a: '%.3g'%(v/n[k][a]) for a,v in c.items()})
PY
