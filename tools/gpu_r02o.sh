#!/bin/bash
# randomised parity on the GPU with pair de-duplication forced on for every (small) scenario, several slices
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r02o; mkdir -p $O
OCT_PHMM_DEDUP=1 OCT_PHMM_SLICES=3 timeout 900 python tools/gpu_fuzz.py 20 50 7000 > $O/fuzz_dedup.log 2>&1; echo "fuzz rc=$?" > $O/rc.log
cat $O/rc.log; tail -3 $O/fuzz_dedup.log
