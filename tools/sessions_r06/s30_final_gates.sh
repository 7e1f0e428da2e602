# Round 6, session 30: the gates once more on the tree that ships (host-side changes since session 27: bench.py's last line, OCT_PHMM_FUSE_TABLES retired): GPU suite, smoke, the default bench line.
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_s30; mkdir -p $O
OCT_PHMM_ENV_SWITCHES=1 timeout 2700 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 5 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.json | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']), j['unit'], round(j['ms_per_step'],2), 'ms', 'verified', j['verified_rows'], j['verified_max_abs_diff'], 'frac', round(j['roofline']['frac'],4), 'stamp', j['roofline']['pmc_summary_matches_these_kernels'])"
