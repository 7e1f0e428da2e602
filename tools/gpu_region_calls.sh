# Region-sized calls on the final build: per-call latency breakdown (Python binding) and the C++ region-call bench (C ABI, region server)
#   bash tools/gpu_region_calls.sh <tag>   -> gpurun_out/<tag>/region_calls.log
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/${1:-region_calls}; mkdir -p $O
{
timeout 200 python tools/latency_breakdown.py
g++ -O2 -std=c++17 tools/region_calls_bench.cpp -o tools/region_calls_bench -Iinclude -Loctopus_amd -loct_phmm -Wl,-rpath,/root/repo/octopus_amd -lpthread 2>&1 | tail -3
timeout 200 ./tools/region_calls_bench 2000 300 24 1 4 8 16 2>&1 | tail -14
} | tee $O/region_calls.log
