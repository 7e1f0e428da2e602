cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r02o; mkdir -p $O
P="--no-small-batch --no-cpu-baseline --no-extras"
for S in 3 4 6 8; do
  OCT_PHMM_SLICES=$S timeout 200 python bench.py $P > $O/bench_slices_$S.json 2>/dev/null
  OCT_PHMM_SLICES=$S timeout 200 python bench.py $P --workload stream > $O/bench_stream_slices_$S.json 2>/dev/null
done
for f in $O/bench*.json; do echo $f $(python -c "
import json; b=json.load(open('$f')); print(round(b['ms_per_step'],2), round(b['value']))"); done
timeout 120 python - <<'PY'
import time, numpy as np, json
from octopus_amd import abi, engine, synth
rng=np.random.default_rng(3)
eng=engine.Engine(abi.Config.default(max_indel_error=16))
res={}
for name,(R,H) in {"300x24":(300,24),"1kx64":(1000,64)}.items():
    b=synth.batch_from_regions([synth.make_region(rng,R,H,B=16,positions="none")]); out=np.empty(b.out_size())
    for _ in range(20): eng.populate(b,out=out)
    t0=time.perf_counter()
    for _ in range(200): eng.populate(b,out=out)
    res[name+"_populate_from_host_ms"]=(time.perf_counter()-t0)/200*1e3
print(json.dumps(res))
PY
