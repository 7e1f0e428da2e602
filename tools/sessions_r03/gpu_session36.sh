cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r03_s36; mkdir -p $O
for R in 1 2 3; do
  timeout 200 ./tools/region_calls_bench 3000 300 24 1 16 2>&1 | grep -v "plain calls" | cut -c1-150 >> $O/region_calls.log
done
OCT_PHMM_UPLOAD_PROFILE=1 timeout 100 ./tools/region_calls_bench 500 300 24 1 2> $O/up.log > /dev/null
python - <<'PY'
import json,collections
rows=[json.loads(l)['upload_profile_ms'] for l in open('/root/repo/gpurun_out/r03_s36/up.log') if l.startswith('{"upload_profile_ms"')]
acc=collections.defaultdict(float)
for r in rows[50:]:
    for k,v in r.items(): acc[k.split(' ')[0]]+=v
n=len(rows)-50
print('single-thread uploads', n, {k: round(v/n,4) for k,v in acc.items()})
PY
cat $O/region_calls.log
