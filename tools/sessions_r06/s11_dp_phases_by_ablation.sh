# Round 6, session 11: the DP kernels without one phase each (OCT_DP_ABLATE builds: timing only), single-slice kernel split of the 12.8 M-pair step.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
for V in default abl1 abl2 abl3 abl4 abl5; do
  L=""; [ $V != default ] && L="OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_$V.so"
  env $L bash tools/gpu_kernel_split.sh r06_s11_$V 100kx128 > /dev/null 2>&1
  echo "$V: $(grep 'k_dp<16, true' gpurun_out/r06_s11_$V/split_100kx128.txt | cut -c88-118)  $(grep 'k_dp<16, false' gpurun_out/r06_s11_$V/split_100kx128.txt | cut -c88-118)  $(grep 'k_walk' gpurun_out/r06_s11_$V/split_100kx128.txt | cut -c88-118)"
done
