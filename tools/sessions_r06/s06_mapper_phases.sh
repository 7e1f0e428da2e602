# Round 6, session 6: where the lane mapper's time goes - timing-only builds: 1 = no counting path, 2 = staging + probes (no pass, no counting), 3 = staging alone.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
for V in default mp1 mp2 mp3; do
  L=""; [ $V != default ] && L="OCT_PHMM_LIB=/root/repo/octopus_amd/variants/liboct_phmm_$V.so"
  env $L bash tools/gpu_kernel_split.sh r06_s06_$V 100kx128 stream-hq > /dev/null 2>&1
  for W in 100kx128 stream-hq; do echo "$V $W: $(grep k_kmer_map_lanes gpurun_out/r06_s06_$V/split_$W.txt | cut -c88-140)  $(grep k_classify gpurun_out/r06_s06_$V/split_$W.txt | cut -c88-140)"; done
done
