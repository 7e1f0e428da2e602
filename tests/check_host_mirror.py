"""The C++ mirror of HaplotypeLikelihoodArray/Model (octopus_amd/host/) through a small driver binary, vs the oracle."""
import subprocess
from pathlib import Path

import numpy as np

import oracle
from backends import ROOT, build_sim
from octopus_amd import abi, engine, synth

SRC = ROOT / "tests" / "host" / "host_mirror_main.cpp"


def build_driver(backend: str) -> Path:
    if backend == "sim":
        lib_dir, lib, exe = build_sim().parent, "phmm_sim", ROOT / "tests" / "host" / "host_mirror_sim"
    else:
        lib_dir, lib, exe = engine.LIB_PATH.parent, "oct_phmm", ROOT / "tests" / "host" / "host_mirror_gpu"
    hdr = ROOT / "octopus_amd" / "host" / "haplotype_likelihood_array.hpp"
    if not exe.exists() or exe.stat().st_mtime < max(SRC.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["g++", "-std=c++17", "-O1", str(SRC), "-o", str(exe), f"-L{lib_dir}", f"-l{lib}", f"-Wl,-rpath,{lib_dir}"], check=True)
    return exe


def scenario(seed, band=8, templates=True, flank=(15, 15), short=False):
    rng = np.random.default_rng(seed)
    g = synth.make_region(rng, 14, 3, T=40, Lh=120, B=band, flank=flank, positions="none")
    if short:       # haplotypes that cannot hold a 40-base read with its two pads
        g["haps"] = [h[:40 + 2 * band - 3] for h in g["haps"]]
        g["begin"] = np.minimum(g["begin"], 5)
    rows, r = [0], 0
    while r < 14:
        r += 2 if (templates and 14 - r >= 2 and rng.random() < 0.5) else 1
        rows.append(r)
    split = len(rows) // 2          # two samples
    return g, rows, split


def run_driver(exe, g, rows, split, band, use_mapq, templates):
    lines = [f"cfg {band} {int(use_mapq)} {int(g['flank'] is not None)} {g['flank'][0] if g['flank'] else 0} {g['flank'][1] if g['flank'] else 0} {int(templates)}",
             f"H {len(g['haps'])}"]
    lines += [f"0 {bytes(h).decode()}" for h in g["haps"]]
    samples = [("S1", rows[:split + 1]), ("S2", rows[split:])]
    lines.append(f"S {len(samples)}")
    for name, rr in samples:
        lines.append(f"{name} {len(rr) - 1}")
        for a, b in zip(rr[:-1], rr[1:]):
            lines.append(str(b - a))
            for r in range(a, b):
                lines.append(f"{int(g['begin'][r])} {int(g['reverse'][r])} {int(g['mapq'][r])} {bytes(g['reads'][r]).decode()} "
                             + ",".join(str(int(q)) for q in g["quals"][r]))
    out = subprocess.run([str(exe)], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout
    return out.strip().split("\n")


def oracle_matrix(g, rows, band, use_mapq):
    haps = []
    for h in g["haps"]:
        n = len(h)
        haps.append(dict(seq=bytes(h), begin=0, gap_open=np.full(n, 45, np.int8), gap_extend=np.full(n, 3, np.int8),
                         mask_fwd=h, prior_fwd=np.full(n, 100, np.int8), mask_rev=h, prior_rev=np.full(n, 100, np.int8)))
    reads = [dict(seq=bytes(g["reads"][r]), quals=g["quals"][r], mapq=int(g["mapq"][r]), reverse=bool(g["reverse"][r]), begin=int(g["begin"][r]))
             for r in range(len(g["reads"]))]
    tmpl = [list(range(a, b)) for a, b in zip(rows[:-1], rows[1:])]
    batch = abi.Batch.from_lists(reads, haps, flank=g["flank"], templates=tmpl)
    cfg = abi.Config.default(max_indel_error=band, use_mapping_quality=int(use_mapq))
    out, st, _ = oracle.populate(cfg, batch)
    return out.reshape(len(haps), len(tmpl)), st


def check(backend, tol=0.0):
    exe = build_driver(backend)
    for seed, band, templates, use_mapq, flank in ((1, 8, True, True, (15, 15)), (2, 16, False, True, None), (3, 8, True, False, (10, 20))):
        g, rows, split = scenario(seed, band, templates, flank)
        if not templates:
            rows = list(range(15))
            split = 7
        lines = run_driver(exe, g, rows, split, band, use_mapq, templates)
        want, st = oracle_matrix(g, rows, band, use_mapq)
        assert st.code == abi.OK
        assert lines[0] == f"pad_requirement {band}"
        got = {}
        for ln in lines:
            if ln.startswith("L "):
                p = ln.split()
                got[(int(p[1]), p[2])] = np.array([float(x) for x in p[3:]])
        for h in range(want.shape[0]):
            row = np.concatenate([got[(h, "S1")], got[(h, "S2")]])
            assert row.shape == want[h].shape and np.max(np.abs(row - want[h])) <= tol, (seed, h, row, want[h])
        # genotype read-out through the mirror of ConstantMixtureGenotypeLikelihoodModel (primed sample S1 = the first `split` rows)
        import itertools
        nh = want.shape[0]
        gts = np.asarray(list(itertools.combinations_with_replacement(range(nh), 2)), np.uint32)
        off = np.arange(nh + 1, dtype=np.uint64) * want.shape[1]
        gl_want = oracle.genotype_likelihoods(want.reshape(-1), off, gts, (0, split))
        gl_got = {(int(p[1]), int(p[2])): float(p[3]) for p in (ln.split() for ln in lines if ln.startswith("G "))}
        for (a, b), w in zip(gts.tolist(), gl_want):
            assert abs(gl_got[(a, b)] - w) <= 1e-9 * max(1.0, abs(w)), (seed, a, b, gl_got[(a, b)], w)
        g3 = [float(ln.split()[1]) for ln in lines if ln.startswith("G3 ")]
        w3 = oracle.genotype_likelihoods(want.reshape(-1), off, np.asarray([[0, 1, 2]], np.uint32), (0, split))[0]
        assert len(g3) == 1 and abs(g3[0] - w3) <= 1e-9 * max(1.0, abs(w3))
        assert "Ghost refused" in lines                       # after reset() there is no device matrix and no host re-implementation
        # realignment through the mirror of HaplotypeLikelihoodModel::align: sample S1's reads against haplotype 0
        n1 = rows[split]
        hap0 = dict(seq=bytes(g["haps"][0]), begin=0, gap_open=np.full(len(g["haps"][0]), 45, np.int8), gap_extend=np.full(len(g["haps"][0]), 3, np.int8),
                    mask_fwd=g["haps"][0], prior_fwd=np.full(len(g["haps"][0]), 100, np.int8), mask_rev=g["haps"][0], prior_rev=np.full(len(g["haps"][0]), 100, np.int8))
        rd = [dict(seq=bytes(g["reads"][r]), quals=g["quals"][r], mapq=int(g["mapq"][r]), reverse=bool(g["reverse"][r]), begin=int(g["begin"][r])) for r in range(n1)]
        ab = abi.Batch.from_lists(rd, [hap0], flank=g["flank"])
        aw, ast = oracle.align_batch(abi.Config.default(max_indel_error=band, use_mapping_quality=int(use_mapq)), ab, 64)
        assert ast.code == abi.OK
        al = [ln.split() for ln in lines if ln.startswith("A ")]
        assert len(al) == n1
        for p in al:
            i = int(p[1])
            assert int(p[2]) == int(aw["mapping_position"][i]) and p[3] == aw["cigar_strings"][i] and abs(float(p[4]) - aw["likelihood"][i]) <= 1e-9, (seed, p, aw["cigar_strings"][i])
        assert any(ln.startswith("primed %d contains 1" % split) for ln in lines)
        assert any(ln == "merged %d" % (len(rows) - 1) for ln in lines)
        assert any(ln == "reset 1 0" for ln in lines)
        assert "extract 1 copy 1 1" in lines                  # extract_sample; a copied array owns its handle, has no device matrix, populates alone
    # error mapping: ShortHaplotypeError and TooLargeBandSizeError surface as the reference's exception types
    g, rows, split = scenario(9, 16, False, None, short=True)
    rows = list(range(15)); split = 7
    lines = run_driver(exe, g, rows, split, 16, True, False)
    assert lines[-1].startswith("ShortHaplotypeError"), lines[-3:]
    _, st = oracle_matrix(g, rows, 16, True)
    assert st.code == abi.ESHORT_HAPLOTYPE and lines[-1] == f"ShortHaplotypeError {st.hap_index} {st.required_extension}"
    lines = run_driver(exe, g, rows, split, 300, True, False)
    assert lines[-1] == "TooLargeBandSizeError 300 256"
