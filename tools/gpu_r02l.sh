cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu -k "mapper or config2 or slices or basic or fuzz or random_scenarios or 100k or server_batches or stream or ragged" > gpurun_out/r02l_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02l_pytest.log
bash tools/gpu_ab.sh r02l OCT_PHMM_LIB=/root/repo/octopus_amd/variants/v3.so
O=gpurun_out/r02l
export OCT_PHMM_SLICES=1
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /root/repo/$O/pmc1 -o p -- python /root/repo/bench.py --no-small-batch --no-cpu-baseline --no-extras --steps 1 --warmup 1 > /dev/null 2>&1)
python - <<'PY'
import csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open('/root/repo/gpurun_out/r02l/pmc1/p_counter_collection.csv')):
    if int(r['Grid_Size'])<1000000: continue
    k=r['Kernel_Name'].split('(')[0][-30:]; agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
for k,c in agg.items():
    if 'kmer_map' in k: print(k, {a: '%.3g'%(v/n[k][a]) for a,v in c.items()})
PY
