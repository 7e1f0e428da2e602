cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r03_s32; mkdir -p $O
OCT_PHMM_UPLOAD_PROFILE=1 timeout 100 ./tools/region_calls_bench 1500 300 24 16 > $O/out.log 2> $O/upload_profile.log
python - <<'PY'
import json,re
rows=[json.loads(l)['upload_profile_ms'] for l in open('/root/repo/gpurun_out/r03_s32/upload_profile.log') if l.startswith('{"upload_profile_ms"')]
import collections
acc=collections.defaultdict(float)
for r in rows:
    for k,v in r.items(): acc[k.split(' ')[0]]+=v
print(len(rows), {k: round(v/len(rows),4) for k,v in acc.items()})
big=[r for r in rows if r['input_MB']>0.3]
acc=collections.defaultdict(float)
for r in big:
    for k,v in r.items(): acc[k]+=v
print('multi-region uploads', len(big), {k: round(v/len(big),4) for k,v in acc.items()})
PY
tail -3 $O/out.log | cut -c1-200
