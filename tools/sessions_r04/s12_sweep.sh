cd /root/repo
for L in 0 25; do echo "#### linger_us=$L"; OCT_PHMM_SERVER_LINGER_US=$L WORKERS="2 3 4 6" CALLERS="16 64 128" bash tools/gpu_server_sweep.sh r04_s12_$L 2>&1 | grep "##\|server"; done
