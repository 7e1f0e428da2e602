# round 3, GPU session 1: device-sized launches on the GPU + baselines for the next steps
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s1; mkdir -p $O
python -c "from octopus_amd import engine; print(engine.kernel_source_sha())" > $O/kernel_source_sha
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "device_sized or populate_basic or templates or random_scenarios or late_traceback or server or chunked or empty or align" > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/rc.log
bash tools/gpu_region_calls.sh r03_s1 > /dev/null 2>&1; echo "region_calls rc=$?" >> $O/rc.log
OCT_PHMM_DEVICE_SIZED=0 timeout 100 python tools/latency_breakdown.py > $O/latency_host_sized.json 2>&1
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/small_trace -o s -- python /root/repo/tools/small_trace.py > /root/repo/$O/small_trace.log 2>&1); echo "small_trace rc=$?" >> $O/rc.log
timeout 120 python tools/long_read_run.py 3 > $O/long_read_before.json 2>$O/long_read_before.err; echo "long rc=$?" >> $O/rc.log
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/long_kstats -o s -- python /root/repo/tools/long_read_run.py 2 > /root/repo/$O/long_kstats.json 2>/root/repo/$O/long_kstats.err); echo "long kstats rc=$?" >> $O/rc.log
find $O -name "*kernel_trace.csv" -size +3M -delete
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" FETCH_SIZE WRITE_SIZE; do
  D=long_pmc_$(echo $C | cut -d' ' -f1)
  (cd /tmp && timeout 150 rocprofv3 --pmc $C --output-format csv -d /root/repo/$O/$D -o p -- python /root/repo/tools/long_read_run.py 2 > /root/repo/$O/$D.json 2> /root/repo/$O/$D.err); echo "$D rc=$?" >> $O/rc.log
done
timeout 120 ./tools/valu_ubench > $O/valu_ubench.log 2>&1; echo "ubench rc=$?" >> $O/rc.log
timeout 200 python tools/dp_clock.py > $O/dp_clock.json 2> $O/dp_clock.err; echo "dp_clock rc=$?" >> $O/rc.log
du -sh $O; cat $O/rc.log; tail -5 $O/pytest_subset.log; cat $O/region_calls.log | cut -c1-600; cat $O/latency_host_sized.json | cut -c1-600; cat $O/long_read_before.json | cut -c1-500; cat $O/dp_clock.json | cut -c1-300; head -12 $O/valu_ubench.log
