cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r03_s13; mkdir -p $O
OCT_PHMM_UPLOAD_PROFILE=1 timeout 600 python tools/stream_e2e.py 1 2 > $O/stream_e2e.json 2> $O/stream_e2e.err
tail -3 $O/stream_e2e.json | cut -c1-900; grep upload_profile $O/stream_e2e.err | sort | uniq -c | sort -rn | head -8
timeout 100 python tools/latency_breakdown.py > $O/latency.json 2>&1; cut -c1-300 $O/latency.json
