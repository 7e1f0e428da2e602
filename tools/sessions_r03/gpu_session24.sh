cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s24; mkdir -p $O
for R in 1 2 3; do
for CFG in "1 0" "0 1" "0 0"; do
  set -- $CFG
  echo "## rep $R merge=$1 fork_early=$2" >> $O/ab.log
  OCT_PHMM_DSL_MERGE_DP=$1 OCT_PHMM_DSL_FORK_EARLY=$2 timeout 200 ./tools/region_calls_bench 3000 300 24 1 16 2>&1 | grep -v "threads\": [48]\|plain calls" | cut -c1-150 >> $O/ab.log
done; done
cat $O/ab.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "device_sized or launch" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
