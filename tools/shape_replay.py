#!/usr/bin/env python3
"""Replay the batch SHAPES of a real Octopus run through the synthetic generator ("capture (R, H, T, Lh) per populate call from the --debug log line at
caller.cpp:1169 once a trace is available").

No chr20 data exists in this container, so every region of bench.py is drawn from SURVEY 8d's distributions (R ~ lognormal(300, 0.8), H ~ geometric(24), Lh 300-500,
flank 40 / 40). The day a trace exists, this tool turns its shapes into the same region files the benches already read, so that the traceback : score-only mix, the
fast-path share and the region server's batch sizes are measured against what the caller really asks:

  python tools/shape_replay.py --csv shapes.csv --out regions.bin          # header R,H,T,Lh[,lhs,rhs]; one row per populate call
  python tools/shape_replay.py --octopus-debug-log octopus_debug.log --out regions.bin [--reads-per-haplotype-median 300] [--read-len 150]
  tools/region_calls_bench --file regions.bin --out results.bin 16 64      # one oct_phmm call per region from 16 / 64 threads (and the region server)

What the reference's debug log gives (ref: src/core/callers/caller.cpp:1168-1171): one line "Calculating likelihoods for <H> haplotypes" per populate call, followed by the
active candidates with their regions. It does not print R, T or Lh: --octopus-debug-log takes H from the line, the active region's width from the candidate lines that
follow where they parse ("<contig>:<begin>-<end>"), Lh = width + 2 x pad (pad = max_indel_error + 15, option_collation.cpp:1715-1719 through min_flank_pad), and draws R
from the lognormal of SURVEY 8d unless --reads-per-haplotype-median says otherwise. A CSV (e.g. from a one-line patch that prints reads.size(), haplotypes.size(), the read
length and the haplotype length beside that debug line) replaces every guess.

Prints a JSON summary: shape quantiles and - from the CPU oracle on a bounded sample - the candidate classes (fast path / score-only DP / traceback DP) of the replayed regions.
"""
from __future__ import annotations

import argparse
import csv
import json
import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from octopus_amd import abi, synth   # noqa: E402

LINE = re.compile(r"Calculating likelihoods for (\d+) haplotypes")
REGION = re.compile(r"\b([\w.]+):(\d+)-(\d+)\b")


def shapes_from_csv(path):
    out = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            r = {k.strip(): v for k, v in row.items()}
            out.append(dict(R=int(r["R"]), H=int(r["H"]), T=int(r.get("T") or 150), Lh=int(r.get("Lh") or 300),
                            lhs=int(r["lhs"]) if r.get("lhs") not in (None, "") else None, rhs=int(r["rhs"]) if r.get("rhs") not in (None, "") else None))
    return out


def shapes_from_debug_log(path, rng, read_len, r_median, band):
    """H per populate call from the reference's own debug line; the active region's width from the first '<contig>:<begin>-<end>' within the next lines."""
    out, lines = [], open(path, errors="replace").read().splitlines()
    pad = band + 15                                          # HaplotypeLikelihoodModel::pad_requirement() + min_flank_pad (ref: haplotype_likelihood_model.cpp:55-58, pair_hmm.hpp:34-38)
    for i, line in enumerate(lines):
        m = LINE.search(line)
        if not m:
            continue
        width = None
        for nxt in lines[i:i + 8]:
            g = REGION.search(nxt)
            if g and int(g.group(3)) > int(g.group(2)):
                width = int(g.group(3)) - int(g.group(2)); break
        Lh = max(read_len + 2 * band + 2, (width if width is not None else 300 - 2 * pad) + 2 * pad)
        R = int(np.clip(rng.lognormal(np.log(r_median), 0.8), 20, 5000))
        out.append(dict(R=R, H=max(1, int(m.group(1))), T=read_len, Lh=Lh, lhs=None, rhs=None))
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--csv"); src.add_argument("--octopus-debug-log")
    ap.add_argument("--out", required=True, help="regions file for tools/region_calls_bench --file / bench.py's region-call legs")
    ap.add_argument("--band", type=int, default=16)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--reads-per-haplotype-median", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--max-regions", type=int, default=0, help="keep the first N shapes (0 = all)")
    ap.add_argument("--classify-sample", type=int, default=16, help="regions the CPU oracle classifies for the summary (0 = none)")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    shapes = shapes_from_csv(a.csv) if a.csv else shapes_from_debug_log(a.octopus_debug_log, rng, a.read_len, a.reads_per_haplotype_median, a.band)
    if a.max_regions:
        shapes = shapes[:a.max_regions]
    if not shapes:
        sys.exit("no populate call found in the input")
    regions = []
    for s in shapes:
        T, Lh = s["T"], max(s["Lh"], s["T"] + 2 * a.band + 2)
        flank = (40, 40) if s["lhs"] is None else (s["lhs"], s["rhs"] or 0)
        flank = (min(flank[0], Lh // 3), min(flank[1], Lh // 3))
        regions.append(synth.make_region(rng, s["R"], s["H"], T=T, Lh=Lh, B=a.band, flank=flank, positions="none"))
    synth.write_regions_file(a.out, regions)
    q = lambda k: [int(x) for x in np.percentile([s[k] for s in shapes], [5, 50, 95, 100])]
    summary = {"regions": len(shapes), "pairs": int(sum(s["R"] * s["H"] for s in shapes)), "out": a.out,
               "quantiles_5_50_95_max": {k: q(k) for k in ("R", "H", "T", "Lh")}}
    if a.classify_sample:
        import oracle
        cfg = abi.Config.default(max_indel_error=a.band)
        tot = {"n_pairs": 0, "n_candidates": 0, "n_fast_path": 0, "n_dp_score_only": 0, "n_dp_traceback": 0}
        for g in regions[:a.classify_sample]:
            _, _, st = oracle.populate(cfg, synth.batch_from_regions([g]))
            for k in tot:
                tot[k] += int(st[k])
        summary["classes_of_the_first_regions"] = dict(tot, regions=min(a.classify_sample, len(regions)),
                                                       traceback_to_score_only=round(tot["n_dp_traceback"] / max(tot["n_dp_score_only"], 1), 3))
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
