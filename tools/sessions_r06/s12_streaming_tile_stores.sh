# Round 6, session 12: the traceback DP's tile stores as non-temporal (streaming) stores, against the shipped library; three interleaved repetitions + kernel split.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s12b; mkdir -p $O
P="--no-small-batch --no-cpu-baseline --no-extras"
for rep in 1 2 3; do for V in default wnt; do
  L=""; [ $V != default ] && L="OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_$V.so"
  env $L timeout 300 python bench.py $P > $O/b_${V}_$rep.json 2> $O/b_${V}_$rep.err
  python -c "
import json; b=json.load(open('$O/b_${V}_$rep.json')); print('$V rep $rep', round(b['ms_per_step'],3), round(b['value'],1), round(b['roofline']['avg_launch_ms'],3), round(b['roofline']['score_only_kernel_avg_launch_ms'],3))"
done; done
OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_wnt.so bash tools/gpu_kernel_split.sh r06_s12_wnt 100kx128 > /dev/null 2>&1
head -8 gpurun_out/r06_s12_wnt/split_100kx128.txt | cut -c1-170
