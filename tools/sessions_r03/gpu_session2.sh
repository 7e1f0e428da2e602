# round 3, GPU session 2: staged walk + fused scan on the GPU; region-call A/B
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s2; mkdir -p $O
python -c "from octopus_amd import engine; print(engine.kernel_source_sha())" > $O/kernel_source_sha
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "device_sized or staged or populate_basic or templates or random_scenarios or late_traceback or server or chunked or empty or align or int32 or wide" > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/rc.log
g++ -O2 -std=c++17 tools/region_calls_bench.cpp -o tools/region_calls_bench -Iinclude -Loctopus_amd -loct_phmm -Wl,-rpath,/root/repo/octopus_amd -lpthread 2>&1 | tail -3
for V in "A=1" "OCT_PHMM_WALK_STAGE=0" "OCT_PHMM_DEVICE_SIZED=0 OCT_PHMM_WALK_STAGE=0" "A=2"; do
  echo "## $V" >> $O/region_calls_ab.log
  env $V timeout 200 ./tools/region_calls_bench 2000 300 24 1 16 2>&1 | tail -4 >> $O/region_calls_ab.log
done
timeout 100 python tools/latency_breakdown.py > $O/latency.json 2>&1
OCT_PHMM_WALK_STAGE=0 timeout 100 python tools/latency_breakdown.py > $O/latency_unstaged.json 2>&1
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/small_trace -o s -- python /root/repo/tools/small_trace.py > /root/repo/$O/small_trace.log 2>&1); echo "small_trace rc=$?" >> $O/rc.log
timeout 200 ./tools/valu_ubench > $O/valu_ubench.log 2>&1; echo "ubench rc=$?" >> $O/rc.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
du -sh $O; cat $O/rc.log; tail -5 $O/pytest_subset.log; cat $O/region_calls_ab.log | cut -c1-300; cat $O/latency.json $O/latency_unstaged.json | cut -c1-400; head -16 $O/valu_ubench.log | cut -c1-260; cut -c1-1500 $O/bench.json
