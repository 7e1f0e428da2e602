#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (p_counter_collection.csv) per kernel of the large-batch run into JSON + markdown.

    python tools/summarize_pmc.py gpurun_out/p5/pmc1 gpurun_out/p5/pmc2 ... --out profiles/r01_step4_pmc_summary

FETCH_SIZE / WRITE_SIZE are in KiB-ish units of 1024 B... rocprofv3 reports them in kilobytes; on gfx950 FETCH_SIZE counts
half of a wide coalesced stream (MI355X_MICROARCH.md §HBM) so `hbm_read_bytes_corrected` doubles it.
"""
import argparse
import collections
import csv
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from octopus_amd import engine   # noqa: E402  (kernel_source_sha only)

ap = argparse.ArgumentParser()
ap.add_argument("passes", nargs="+")
ap.add_argument("--out", required=True)
ap.add_argument("--min-grid", type=int, default=1_000_000)
ap.add_argument("--sha", default=None, help="kernel_source_sha recorded on the GPU box beside the passes (default: this tree's)")
a = ap.parse_args()
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(list)
def rows_of(p):
    """One dict per (dispatch, counter): rocprofv3's CSV output, or its default rocpd database (p_results.db)."""
    f = Path(p) / "p_counter_collection.csv"
    if f.exists():
        yield from csv.DictReader(open(f))
        return
    import sqlite3
    db = sqlite3.connect(str(Path(p) / "p_results.db"))
    q = ("select grid_size, kernel_name, counter_name, sum(value), min(start), max(end) from counters_collection "
         "group by dispatch_id, counter_name")
    for g, k, n, v, t0, t1 in db.execute(q):
        yield {"Grid_Size": g, "Kernel_Name": k, "Counter_Name": n, "Counter_Value": v, "Start_Timestamp": t0, "End_Timestamp": t1}


for p in a.passes:
    for r in rows_of(p):
        g = int(r["Grid_Size"])
        if g < a.min_grid:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, c in agg.items():
    d = {n: v / calls[k][n] for n, v in c.items()}        # per launch
    d["launch_ns_under_pmc"] = sum(dur[k]) / len(dur[k])
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes_corrected"] = d["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in d:
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
    if "SQ_INSTS_VALU" in d and "SQ_ACTIVE_INST_VALU" in d:
        d["valu_quad_cycles_per_inst"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_INSTS_VALU"]
    out[k] = d
# stamp: which kernels these counters belong to. bench.py compares kernel_source_sha with the tree it runs in (the GPU box has no .git)
# and flags a summary collected on other kernels; the git hash is for humans (run this script at the commit that was profiled).
git = subprocess.run(["git", "-C", str(ROOT), "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "-C", str(ROOT), "status", "--porcelain", "--", "octopus_amd/csrc", "include"], capture_output=True, text=True).stdout.strip())
out["_meta"] = {"kernel_source_sha": a.sha or engine.kernel_source_sha(), "git": git + ("+dirty" if dirty else ""), "passes": [str(p) for p in a.passes]}
Path(a.out + ".json").write_text(json.dumps(out, indent=1) + "\n")
cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
        "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "launch_ns_under_pmc"]
with open(a.out + ".md", "w") as f:
    f.write("| kernel | " + " | ".join(cols) + " |\n|---|" + "---|" * len(cols) + "\n")
    for k, d in sorted(out.items()):
        if k == "_meta":
            continue
        f.write(f"| {k} | " + " | ".join(f"{d.get(c, float('nan')):.4g}" for c in cols) + " |\n")
print(json.dumps({k: {n: d[n] for n in d if n.startswith("hbm") or n.startswith("valu") or n == "GRBM_GUI_ACTIVE"} for k, d in out.items() if k != "_meta"}, indent=1))
