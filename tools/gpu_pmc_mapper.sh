# PMC passes with the instruction-mix and wait counters (all kernels of a single-slice run): where do the issue cycles go?
set -x
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/${1:-pmcmap}; mkdir -p $O
export OCT_PHMM_SLICES=1
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /root/repo/$O/pmc_a -o p -- python /root/repo/bench.py --no-small-batch --no-cpu-baseline --steps 1 --warmup 1 > /root/repo/$O/pmc_a.json 2> /root/repo/$O/pmc_a.err); echo "pmc_a rc=$?" >> $O/rc.log
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d /root/repo/$O/pmc_b -o p -- python /root/repo/bench.py --no-small-batch --no-cpu-baseline --steps 1 --warmup 1 > /root/repo/$O/pmc_b.json 2> /root/repo/$O/pmc_b.err); echo "pmc_b rc=$?" >> $O/rc.log
cat $O/rc.log; tail -3 $O/pmc_a.err $O/pmc_b.err; ls -la $O/pmc_a $O/pmc_b
python - <<'PY'
import csv,collections,glob,sys
for d in ("pmc_a","pmc_b"):
    for f in glob.glob(f"/root/repo/gpurun_out/*/{d}/**/p_counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:40]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        for k,v in agg.items():
            if max(v.values())>1e8: print(k, {c: f"{x:.3g}" for c,x in v.items()})
PY
