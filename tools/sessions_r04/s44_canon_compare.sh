cd /root/repo; export OCT_PHMM_ENV_SWITCHES=1 OCT_PHMM_DEDUP=1 OCT_DEBUG_CANON=1
O=gpurun_out/r04_s44; mkdir -p $O
OCT_DEBUG_CANON_FILE=/tmp/canon_lds.bin timeout -k 5 300 python tools/sessions_r04/s43_canon_debug.py > /dev/null 2>&1
OCT_PHMM_WINDOW_LDS=0 OCT_DEBUG_CANON_FILE=/tmp/canon_glob.bin timeout -k 5 300 python tools/sessions_r04/s43_canon_debug.py > /dev/null 2>&1
python - <<'PY' > $O/cmp.txt
import numpy as np
a = np.fromfile("/tmp/canon_lds.bin", dtype=np.uint32); b = np.fromfile("/tmp/canon_glob.bin", dtype=np.uint32)
x = np.arange(len(a), dtype=np.uint32)
lo, hi = 246184, 331896
bad = np.nonzero((a[lo:hi] < lo) | (a[lo:hi] > x[lo:hi]))[0] + lo
print("bad", len(bad), "values:", np.unique(a[bad])[:10])
print("golden for the bad ones: self", int(np.sum(b[bad] == bad)), "other", int(np.sum(b[bad] != bad)))
print("first 20 bad x - lo:", (bad[:20] - lo).tolist()); print("golden - lo:", (b[bad[:20]].astype(np.int64) - lo).tolist())
diff = np.nonzero(a != b)[0]; print("differences overall", len(diff), "of which bad", int(np.isin(diff, bad).sum()))
good = np.setdiff1d(np.arange(lo, hi), bad)
print("good windows of the region: golden self", int(np.sum(b[good] == good)), "other", int(np.sum(b[good] != good)))
# haplotype of the bad windows
print("bad (x - lo) // 487 unique count", len(np.unique((bad - lo) // 487)), "max", int(((bad - lo) // 487).max()))
print("bad (x - lo) % 487 hist head", np.bincount((bad - lo) % 487)[:40].tolist())
PY
cat $O/cmp.txt
