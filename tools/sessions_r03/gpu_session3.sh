# round 3, GPU session 3: why do device-sized launches hurt multi-region (server) batches?
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s3; mkdir -p $O
g++ -O2 -std=c++17 tools/region_calls_bench.cpp -o tools/region_calls_bench -Iinclude -Loctopus_amd -loct_phmm -Wl,-rpath,/root/repo/octopus_amd -lpthread 2>&1 | tail -3
for N in 1 4 8; do
  for V in "A=1" "OCT_PHMM_DEVICE_SIZED=0" "OCT_PHMM_WALK_STAGE=0"; do
    echo "## regions=$N $V" >> $O/multi_region.log
    env $V timeout 100 python tools/multi_region_trace.py $N >> $O/multi_region.log 2>&1
  done
done
for V in "A=1" "OCT_PHMM_DEVICE_SIZED=0"; do
  echo "## $V" >> $O/server_profile.log
  env $V OCT_PHMM_SERVER_PROFILE=1 timeout 200 ./tools/region_calls_bench 2000 300 24 1 16 2>&1 | grep -v "handle per thread" | tail -6 >> $O/server_profile.log
done
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace4 -o s -- python /root/repo/tools/multi_region_trace.py 4 > /root/repo/$O/trace4.log 2>&1); echo "trace4 rc=$?" >> $O/rc.log
(cd /tmp && OCT_PHMM_DEVICE_SIZED=0 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace4_host -o s -- python /root/repo/tools/multi_region_trace.py 4 > /root/repo/$O/trace4_host.log 2>&1); echo "trace4_host rc=$?" >> $O/rc.log
timeout 100 python tools/latency_breakdown.py > $O/latency.json 2>&1
timeout 200 ./tools/valu_ubench > $O/valu_ubench.log 2>&1; echo "ubench rc=$?" >> $O/rc.log
cat $O/rc.log; cat $O/multi_region.log | cut -c1-260; cat $O/server_profile.log | cut -c1-400; cut -c1-420 $O/latency.json; head -50 $O/valu_ubench.log | cut -c1-250
