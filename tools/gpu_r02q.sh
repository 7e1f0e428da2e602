cd /root/repo
timeout 200 python tools/latency_breakdown.py
g++ -O2 -std=c++17 tools/region_calls_bench.cpp -o tools/region_calls_bench -Iinclude -Loctopus_amd -loct_phmm -Wl,-rpath,/root/repo/octopus_amd -lpthread 2>&1 | tail -3
timeout 120 ./tools/region_calls_bench 2>&1 | tail -12
