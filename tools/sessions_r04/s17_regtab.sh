cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r04_s17; mkdir -p $O
timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -x -q -k "shared or basic or fuzz or random" > $O/pytest_subset.log 2>&1; echo "parity subset rc=$?"; tail -2 $O/pytest_subset.log
bash tools/gpu_kernel_split.sh r04_s17 stream 2>&1 | grep "##\|k_window\|k_hap_tables\|k_dedup"
timeout -k 5 300 python tools/stream_e2e.py 2 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print({k: (round(v['x_resident'],3), round(v['ms_per_batch'],2)) for k,v in d.items() if k.startswith('in_flight')}, d['resident_ms'], d['split_ms'])"
