cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s36; mkdir -p $O
timeout -k 5 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.log
timeout -k 5 300 python tools/long_read_legs.py > $O/legs.json 2> $O/legs.err; echo "legs rc=$?" >> $O/rc.log
timeout -k 5 200 python tools/latency_breakdown.py > $O/lat.json 2> $O/lat.err
cat $O/rc.log; tail -3 $O/pytest.log; cut -c1-200 $O/legs.json; cut -c1-400 $O/lat.json
