# Round 6, session 24: counters for the long-read kernels (k_kmer_map_big, k_dp_rows, k_walk_rows) on ccs2048x12 and ccs256x12: where do the cycles go?
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s24; mkdir -p $O
(cd /tmp && timeout 60 rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TA_[A-Z_0-9a-z]*\|TCP_[A-Z_0-9a-z]*\|TCC_[A-Z_0-9a-z]*" | sort -u | tr '\n' ' ' > /root/repo/$O/counters_avail.txt)
wc -c $O/counters_avail.txt
pass() { n=$1; leg=$2; shift 2
  (cd /tmp && timeout 200 rocprofv3 --pmc "$@" --output-format csv -d /root/repo/$O/pmc_${leg}_$n -o p -- python /root/repo/tools/long_read_legs.py $leg > /dev/null 2> /root/repo/$O/pmc_${leg}_$n.err); echo "pass $n $leg rc=$?"; }
for leg in ccs2048x12 ccs256x12; do
pass a $leg SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass b $leg SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA
pass c $leg SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
pass d $leg SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN
pass e $leg SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_WAVES_EQ_64
pass f $leg GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE
done
python - <<'PY' > gpurun_out/r06_s24/summary.txt
import csv,collections,glob
for leg in ("ccs2048x12","ccs256x12"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.defaultdict(set)
    for f in glob.glob(f"/root/repo/gpurun_out/r06_s24/pmc_{leg}_*/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("octphmm::","").replace("void ","")[:40]
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); calls[(k,r["Counter_Name"])].add(r.get("Dispatch_Id",""))
    print("##", leg, "(sums over all launches of the run: 5 steps)")
    for k,v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES",0)):
        if v.get("SQ_BUSY_CYCLES",0) < 1e6 and v.get("SQ_WAVE_CYCLES",0) < 1e7: continue
        print(k); print("   ", {c: f"{x:.4g}" for c,x in sorted(v.items())}, "launches", len(calls[(k,"SQ_WAVES")]))
PY
cat $O/summary.txt | cut -c1-900
find $O -name "*counter_collection.csv" -size +2M -delete
