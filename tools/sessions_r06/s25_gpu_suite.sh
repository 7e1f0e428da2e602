# Round 6, session 25: the GPU test suite and a fuzz subset on the round's last kernels, then the long-read and region-sized legs once more.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s25; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
timeout -k 5 900 python tools/gpu_fuzz.py shapes 1000 31000 12 > $O/gpu_fuzz_shapes_1000.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/gpu_fuzz_shapes_1000.log | cut -c1-300
timeout 300 python tools/long_read_legs.py 2>/dev/null | cut -c1-100
for N in 1 4 16; do timeout 120 python tools/mid_batch_trace.py $N 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$N regions', j['ms'])"; done
