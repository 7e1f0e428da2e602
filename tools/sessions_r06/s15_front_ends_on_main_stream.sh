# Round 6, session 15: the front ends of all slices on the handle's high-priority stream, every slice's phase 2 on a normal-priority stream of its own (OCT_PHMM_P1_MAIN=1) against
# the shipped schedule (a slice's phase 1 and 2 on one stream); interleaved; then a verified run and the multi-region GPU checks with it on.
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r06_s15; mkdir -p $O
P="--no-small-batch --no-cpu-baseline --no-extras"
for rep in 1 2 3; do for M in 0 1; do for W in 100kx128 stream stream-hq; do
  OCT_PHMM_P1_MAIN=$M timeout 300 python bench.py $P --workload $W > $O/b_${M}_${W}_$rep.json 2> $O/b_${M}_${W}_$rep.err
  python -c "
import json; b=json.load(open('$O/b_${M}_${W}_$rep.json')); print('p1_main $M $W rep $rep', round(b['ms_per_step'],3), round(b['value'],1))"
done; done; done
for S in 6 8; do OCT_PHMM_SLICES=$S OCT_PHMM_P1_MAIN=1 timeout 300 python bench.py $P > $O/b_1_slices$S.json 2>/dev/null; python -c "
import json; b=json.load(open('$O/b_1_slices$S.json')); print('p1_main 1 slices $S', round(b['ms_per_step'],3), round(b['value'],1))"; done
OCT_PHMM_P1_MAIN=1 timeout 600 python bench.py --no-small-batch --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_verified.json 2> $O/bench_verified.err
python -c "
import json; b=json.load(open('$O/bench_verified.json')); print({k:b.get(k) for k in ('value','ms_per_step','verified_rows','verified_max_abs_diff')}, {k:(b[k]['ms'], b[k]['verified_max_abs_diff']) for k in ('stream','stream_hq','hq')}, b['stream'].get('e2e_pipelined_2_vs_resident'), b['stream'].get('e2e_pipelined_3_vs_resident'))"
OCT_PHMM_P1_MAIN=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu > $O/gpu_fullsize.log 2>&1; echo "fullsize rc=$?"; tail -2 $O/gpu_fullsize.log
