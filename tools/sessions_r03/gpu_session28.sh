cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s28; mkdir -p $O
for R in 1 2; do for W in 0 1 2; do for F in 0 1; do
  echo "## rep $R WALK_STAGE=$W FORK_EARLY=$F" >> $O/small.log
  OCT_PHMM_WALK_STAGE=$W OCT_PHMM_DSL_FORK_EARLY=$F OCT_LAT_SMALL=1 timeout 100 python tools/latency_breakdown.py 2>&1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:round(v['run']+v['wait'],4) for k,v in d.items()})" >> $O/small.log
done; done; done
for R in 1 2 3; do for W in 2 3; do
  echo "## rep $R workers=$W" >> $O/server.log
  OCT_PHMM_SERVER_WORKERS=$W timeout 200 ./tools/region_calls_bench 3000 300 24 8 16 2>&1 | grep '"server"' | cut -c1-150 >> $O/server.log
done; done
cat $O/small.log $O/server.log
