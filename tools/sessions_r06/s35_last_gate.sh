# Round 6, session 35: the GPU suite as it ships (with the align check on 33 k-base reads added after the last profile set) and smoke, once more.
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06_s35; mkdir -p $O
OCT_PHMM_ENV_SWITCHES=1 timeout 2700 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-120
