# Round 5, GPU session 15: kernel launches per device batch of the region server under load (64 callers, the stream's regions): rocprofv3 --kernel-trace --stats of region_calls_bench
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r05_s15; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from octopus_amd import synth
synth.write_regions_file("/tmp/stream_regions.bin", synth.region_stream_shard(seed=42, n_regions=2000, B=16, positions="none"))
PY
(cd /tmp && OCT_PHMM_SERVER_PROFILE=1 OCT_BENCH_REPS=2 timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof -o s -- /root/repo/tools/region_calls_bench --file /tmp/stream_regions.bin 64 > /root/repo/$O/server_under_rocprof.log 2>&1)
python - $O <<'PY' | tee $O/launches_per_batch.txt
import csv, glob, json, re, sys
o = sys.argv[1]
log = open(o + "/server_under_rocprof.log").read()
prof = json.loads(re.search(r'\{"server_profile_ms".*\}', log).group(0))["server_profile_ms"]
rate = json.loads(re.search(r'\{"mode": "server".*\}', log).group(0))
rows = list(csv.DictReader(open(glob.glob(o + "/prof/**/*kernel_stats.csv", recursive=True)[0])))
batches = prof["batches"]
print(f"region server under rocprofv3 --kernel-trace, 64 callers: {rate['regions_per_s']:.0f} regions/s, {rate['regions_per_device_batch']:.1f} regions per device batch; {prof['calls']} calls in {batches} device batches (server batches only; the bench's 64 + 2,000 plain verification calls launch their own chains: 10 kernels each)")
tot = 0
for r in rows:
    n = int(r["Calls"]); tot += n
    print(f"  {r['Name'][:70]:70s} {n:7d} launches  {n / batches:6.2f} per batch  avg {float(r['AverageNs']) / 1e3:8.1f} us")
plain = 64
print(f"total kernel launches {tot}; minus {plain} plain calls x 10 = {tot - 10 * plain}; per device batch {(tot - 10 * plain) / batches:.2f}")
PY
find $O/prof -name "*kernel_trace.csv" -delete
