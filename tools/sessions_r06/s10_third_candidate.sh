# Round 6, session 10: the lane mapper's third candidate diagonal, the pair sort with four loads in flight, pairing only where a haplotype's runs are long (the stream must not pair):
# bench lines of the three workloads, mapper statistics, kernel splits, 16-region timeline with the lane mapper forced / default.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s10; mkdir -p $O
P="--no-small-batch --no-cpu-baseline --no-extras"
for rep in 1 2; do for W in 100kx128 stream stream-hq; do
  timeout 300 python bench.py $P --workload $W > $O/b_${W}_$rep.json 2> $O/b_${W}_$rep.err
  python -c "
import json; b=json.load(open('$O/b_${W}_$rep.json')); print('$W rep $rep', round(b['ms_per_step'],3), round(b['value'],1))"
done; done
OCT_PHMM_MAP_STATS=1 timeout 300 python bench.py $P --steps 1 --warmup 0 2>&1 >/dev/null | grep mapper_pairs | tail -1
bash tools/gpu_kernel_split.sh r06_s10 100kx128 stream-hq stream > /dev/null 2>&1
for f in gpurun_out/r06_s10/split_*.txt; do echo "### $f"; head -14 $f | cut -c1-175; done
for LM in default 1; do
  E=""; [ $LM != default ] && E="OCT_PHMM_LANE_MAPPER=$LM"
  env $E timeout -k 5 120 python tools/mid_batch_trace.py 16 2>&1 | tail -1 | cut -c1-300
  T=/tmp/trace_$LM; rm -rf $T
  (cd /tmp && env $E timeout -k 5 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $T -o s -- python /root/repo/tools/mid_batch_trace.py 16 > /dev/null 2>&1)
  python tools/timeline_tail.py $T > $O/timeline_16_regions_lane_mapper_$LM.txt 2>&1; rm -rf $T
  echo "### timeline lane mapper $LM"; cat $O/timeline_16_regions_lane_mapper_$LM.txt | cut -c1-120
done
