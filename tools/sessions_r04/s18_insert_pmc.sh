cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1 OCT_PHMM_SLICES=1
O=gpurun_out/r04_s18; mkdir -p $O
P="--no-small-batch --no-cpu-baseline --no-extras --workload stream --steps 1 --warmup 1"
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_GDS" "TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA_WRREQ_sum TCC_EA_RDREQ_sum TCC_EA_ATOMIC_sum"; do
  D=pmc_$(echo $C | cut -d' ' -f1)
  (cd /tmp && timeout -k 5 200 rocprofv3 --pmc $C --output-format csv -d /root/repo/$O/$D -o p -- python /root/repo/bench.py $P > /root/repo/$O/$D.json 2> /root/repo/$O/$D.err); echo "$D rc=$?"
done
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r04_s18/pmc_*/**/p_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_window" in k or "k_hap_tables" in k:
            kk = k.split("(")[0].replace("octphmm::", "")
            agg[kk][r["Counter_Name"]] += float(r["Counter_Value"]); n[kk].add(r["Dispatch_Id"])
    for kk, c in sorted(agg.items()):
        print(kk, len(n[kk]), {a: f"{x / len(n[kk]):.3g}" for a, x in sorted(c.items())})
PY
