cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s20; mkdir -p $O
for N in 8 16 64; do
  for V in "" "OCT_PHMM_LANE_MAPPER=1" "OCT_PHMM_DEDUP=1" "OCT_PHMM_WALK_STAGE=2" "OCT_PHMM_WALK_STAGE=1" "OCT_PHMM_DSL_MERGE_DP=1" "OCT_PHMM_DSL_MERGE_DP=0" "OCT_PHMM_DSL_FORK_EARLY=0" "OCT_PHMM_LANE_MAPPER=1 OCT_PHMM_WALK_STAGE=2" "OCT_PHMM_LANE_MAPPER=1 OCT_PHMM_DEDUP=1 OCT_PHMM_WALK_STAGE=2"; do
    echo "N=$N [$V] $(env $V timeout -k 5 100 python tools/mid_batch_trace.py $N 2>&1 | cut -c1-260)" >> $O/sweep.log
  done
done
cat $O/sweep.log
