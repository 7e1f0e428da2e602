# Round 5, GPU session 3: the region server's batching policy (workers x gather) and A/B libraries of the traceback DP (tile stride, read-record chunks).
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r05_s03; mkdir -p $O
timeout -k 5 600 python -m pytest tests -m gpu -x -q -k "server or shape or launch_modes or late or host_mirror or patched" 2>&1 | tail -6 > $O/gpu_tests_subset.log
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
{
for W in 1 2 3; do for G in 1 0; do
  echo "## server [workers $W gather $G]"; env OCT_PHMM_SERVER_WORKERS=$W OCT_PHMM_SERVER_GATHER=$G OCT_BENCH_REPS=3 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 16 64 128 2>&1 | grep "\"server\|differ"
done; done
echo "## server [workers 2 no pipeline]"; env OCT_PHMM_SERVER_WORKERS=2 OCT_PHMM_SERVER_PIPELINE=0 OCT_BENCH_REPS=3 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 16 64 128 2>&1 | grep "\"server\|differ"
echo "## server [workers 2 profile]"; env OCT_PHMM_SERVER_WORKERS=2 OCT_PHMM_SERVER_PROFILE=1 OCT_BENCH_REPS=3 timeout -k 5 200 ./tools/region_calls_bench --file /tmp/stream_regions.bin 64 2>&1 | grep "\"server\|differ\|profile"
echo "## 300x24 regions"; timeout -k 5 200 ./tools/region_calls_bench 3000 300 24 1 16 64 2>&1 | grep "server\|handle per"
} > $O/server_sweep.log 2>&1
{
for SW in "OCT_PHMM_DEVICE_SIZED=1" "OCT_PHMM_DEVICE_SIZED=0"; do echo "## upload profile 16 regions [$SW]"; env $SW OCT_PHMM_UPLOAD_PROFILE=1 timeout -k 5 100 python tools/mid_batch_trace.py 16 2>&1 | grep upload_profile | tail -3; done
} > $O/upload_profile.log 2>&1
P="--no-small-batch --no-cpu-baseline --no-extras"
{
for V in "" "OCT_PHMM_REC_CHUNK=64" "OCT_PHMM_REC_CHUNK=96" "OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_stride65.so" "OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_stride68.so"; do
  echo "## headline [$V]"; env $V timeout -k 5 200 python bench.py $P 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k:round(b[k],3) for k in ('value','ms_per_step')}, 'trace', round(b['roofline']['avg_launch_ms'],3), 'score', round(b['roofline']['score_only_kernel_avg_launch_ms'],3))"
  echo "## stream [$V]"; env $V timeout -k 5 200 python bench.py $P --workload stream 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k:round(b[k],3) for k in ('value','ms_per_step')})"
done
} > $O/dp_variants.log 2>&1
export OCT_PHMM_SLICES=1
for V in default stride65 stride68; do
  L=""; [ $V != default ] && L="OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_$V.so"
  (cd /tmp && env $L timeout -k 5 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d /root/repo/$O/pmc_$V -o p -- python /root/repo/bench.py $P --steps 2 --warmup 1 > /dev/null 2>&1)
  python - $O/pmc_$V $V <<'PY' >> $O/lds_conflicts.log 2>&1
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("octphmm::", "").replace("void ", "")[:40]
        if "k_dp" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, c in acc.items():
    print(sys.argv[2], k, {name: f"{v / max(1, n[(k, name)]):.4g}" for name, v in sorted(c.items())})
PY
  rm -rf $O/pmc_$V
done
unset OCT_PHMM_SLICES
tail -3 $O/gpu_tests_subset.log; cat $O/server_sweep.log $O/upload_profile.log $O/dp_variants.log $O/lds_conflicts.log | cut -c1-300
