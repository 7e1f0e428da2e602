cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s16; mkdir -p $O
for W in 2 3 4 2 3; do
  echo "## workers=$W" >> $O/server_workers.log
  OCT_PHMM_SERVER_WORKERS=$W timeout 200 ./tools/region_calls_bench 3000 300 24 8 16 2>&1 | grep '"server"' >> $O/server_workers.log
done
cat $O/server_workers.log | cut -c1-170
