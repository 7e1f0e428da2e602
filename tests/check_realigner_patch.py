"""INTEGRATION.md's third seam, compiled: the reference's OWN realignment of the reads assigned to a haplotype (src/core/tools/read_realigner.cpp:83-155 - k-mer
table, model.reset, model.align read by read, AlignedRead::realign; cut out of a copy of the file by oracle/make_patched_tree.py and compiled between stand-in
types in oracle/ref_realigner_bridge.cpp) against the same functions with the last one replaced by integration/read_realigner_on_device.inc
(reset -> pack -> ONE oct_phmm_align with the device's k-mer mapper -> AlignedRead::realign). Both sides run the reference's real repeat-based error models on
the haplotype. Compared per read: the new region (begin, end), the CIGAR operation by operation, the log-likelihood. Reads with substitutions, insertions and
deletions against the haplotype, with and without the log-likelihood vector, mapping quality on / capped / off, bands 8 - 32; a haplotype too short for its reads
(both sides: ShortHaplotypeError with the same required extension)."""
import ctypes as C
import json
from pathlib import Path

import numpy as np

from check_assigner_patch import default_tables

ROOT = Path(__file__).resolve().parents[1]
_LIBS = {"ref": "libref_realigner.so", "patched_sim": "libref_realigner_patched_sim.so", "patched_gpu": "libref_realigner_patched_gpu.so"}
_loaded = {}
MAX_OPS = 96


def have(which: str) -> bool:
    return (ROOT / "oracle" / "_ref" / _LIBS[which]).exists()


def _lib(which):
    if which not in _loaded:
        _loaded[which] = C.CDLL(str(ROOT / "oracle" / "_ref" / _LIBS[which]))
    return _loaded[which]


class _Args(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("max_indel_error", "use_int_scores", "use_mapping_quality", "mapping_quality_cap", "mapping_quality_cap_trigger")] + [
        ("tables", C.c_void_p), ("table_lens", C.c_void_p), ("hap_bases", C.c_void_p), ("hap_len", C.c_uint32), ("hap_begin", C.c_int64),
        ("n_reads", C.c_uint32), ("read_bases", C.c_void_p), ("quals", C.c_void_p), ("read_off", C.c_void_p), ("read_begin", C.c_void_p),
        ("mapq", C.c_void_p), ("reverse", C.c_void_p), ("want_likelihoods", C.c_int32)]


def scenario(rng, n_reads, T, Lh, band, short=False):
    """One haplotype (with a homopolymer and a dinucleotide run for the error models) and reads drawn from it: substitutions at the base qualities' rates,
    every third read with a 1-4 base insertion or deletion, read positions off by up to two bases (the mapper has to find them)."""
    acgt = np.frombuffer(b"ACGT", np.uint8)
    hap = acgt[rng.integers(0, 4, Lh)]
    hap[Lh // 3:Lh // 3 + 12] = ord("A")
    hap[Lh // 2:Lh // 2 + 16] = np.frombuffer(b"CA" * 8, np.uint8)
    start = 20_000
    reads, quals, rbegin = [], [], []
    for r in range(n_reads):
        o = int(rng.integers(band + 3, Lh - T - band - 8))
        s = hap[o:o + T + 6].copy()
        if r % 3 == 1:
            p = int(rng.integers(10, T - 10)); n = int(rng.integers(1, 5))
            s = np.concatenate([s[:p], acgt[rng.integers(0, 4, n)], s[p:]])
        elif r % 3 == 2:
            p = int(rng.integers(10, T - 10)); n = int(rng.integers(1, 5))
            s = np.concatenate([s[:p], s[p + n:]])
        s = s[:T].copy()
        q = rng.choice(np.asarray([12, 25, 37], np.uint8), size=T, p=[0.1, 0.2, 0.7])
        flip = rng.random(T) < np.where(q == 37, 0.002, np.where(q == 25, 0.01, 0.08))
        s[flip] = acgt[rng.integers(0, 4, int(flip.sum()))]
        reads.append(s); quals.append(q); rbegin.append(start + o + int(rng.integers(-2, 3)))
    if short:                                                          # a haplotype that cannot hold a read with its two pads anywhere (haplotype_likelihood_model.cpp:378-391)
        hap = hap[:T + 2 * band - 3]
        rbegin = [start + band - 1] * n_reads
    return dict(hap=hap, hap_begin=start, reads=reads, quals=quals, rbegin=np.asarray(rbegin, np.int64), mapq=rng.integers(0, 70, n_reads).astype(np.uint8),
                reverse=rng.integers(0, 2, n_reads).astype(np.uint8))


def repeat_scenario(rng, band=16, T=60):
    """reads that lie ENTIRELY inside a (CA)n run / a poly-A run longer than themselves. Every 6-mer of such a read occurs at every (second) position of
    the run, so it ties on (run - T) / period + 1 diagonals - 36 and 61 here, more than the 15 slots of the ABI: the reference maps realigned reads without a cap
    (read_realigner.cpp:128,137) and model.align sees all of them. A few carry one substitution (so the DP, not the exact-match shortcut, decides among the ties),
    a few start at the run's far end (where a capped, ascending candidate list never reaches), and ordinary reads sit beside them."""
    acgt = np.frombuffer(b"ACGT", np.uint8)
    Lh = 520
    hap = acgt[rng.integers(0, 4, Lh)]
    ca0, ca_n, a0, a_n = 70, 130, 290, 120
    hap[ca0:ca0 + ca_n] = np.frombuffer(b"CA" * (ca_n // 2), np.uint8)
    hap[a0:a0 + a_n] = ord("A")
    hap[a0 - 1] = ord("G"); hap[a0 + a_n] = ord("C"); hap[ca0 - 1] = ord("T"); hap[ca0 + ca_n] = ord("G")
    start = 20_000
    reads, quals, rbegin = [], [], []
    def add(o, sub=None, shift=0):
        s = hap[o:o + T].copy()
        if sub is not None:
            s[sub] = ord("G") if s[sub] != ord("G") else ord("T")
        reads.append(s); quals.append(np.full(T, 30, np.uint8)); rbegin.append(start + o + shift)
    for o in (ca0, ca0 + 2, ca0 + 34, ca0 + ca_n - T, ca0 + ca_n - T - 2):
        add(o); add(o, sub=T // 2); add(o, shift=4)
    for o in (a0, a0 + 1, a0 + 30, a0 + a_n - T):
        add(o); add(o, sub=T // 3, shift=-3)
    for o in (20, 210, 430):                                      # ordinary reads: unique mapping, the device's result stays
        add(o); add(o, sub=7)
    n = len(reads)
    return dict(hap=hap, hap_begin=start, reads=reads, quals=quals, rbegin=np.asarray(rbegin, np.int64), mapq=rng.integers(20, 70, n).astype(np.uint8),
                reverse=rng.integers(0, 2, n).astype(np.uint8))


def realign(which, sc, band, want_ll=True, use_mapq=True, mapq_cap=False, max_ops=None):
    max_ops = max_ops or MAX_OPS
    flat, lens = default_tables()
    cat = lambda xs: (np.concatenate(xs).astype(np.uint8), np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.uint32))
    qb, qo = cat(sc["reads"]); ql, _ = cat(sc["quals"])
    hap = np.ascontiguousarray(sc["hap"])
    n = len(sc["reads"])
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    a = _Args(band, 0, int(use_mapq), 40 if mapq_cap else 255, 30 if mapq_cap else -1, p(flat), p(lens), p(hap), len(hap), int(sc["hap_begin"]),
              n, p(qb), p(ql), p(qo), p(sc["rbegin"]), p(sc["mapq"]), p(sc["reverse"]), int(want_ll))
    begin, end = np.zeros(n, np.int64), np.zeros(n, np.int64)
    n_ops, ops, ll = np.zeros(n, np.uint32), np.zeros((n, max_ops), np.uint32), np.zeros(n)
    ext = C.c_uint32(0)
    rc = _lib(which).ref_realigner_realign(C.byref(a), p(begin), p(end), p(n_ops), p(ops), max_ops, p(ll), C.byref(ext))
    cigars = ["".join(f"{int(w) >> 8}{chr(int(w) & 0xff)}" for w in ops[i, :n_ops[i]]) for i in range(n)]
    return dict(rc=rc, ext=ext.value, begin=begin.tolist(), end=end.tolist(), cigar=cigars, loglik=ll.tolist())


GOLDEN = ROOT / "tests" / "golden" / "realigner_seam_golden.json"
SCENARIOS = ((8, 18, 60, 220, True, True, False), (16, 24, 100, 330, True, True, True), (16, 15, 150, 420, False, False, False), (32, 12, 120, 420, True, False, False))


def check(backend, tol=0.0, golden=False):
    """golden: compare with what the reference's functions produced where tests/golden/make_realigner_seam_golden.py ran (committed) - the GPU box's run, which
    needs only the patched library."""
    lib = "patched_" + backend
    rng = np.random.default_rng(91)
    stored = json.loads(GOLDEN.read_text())["results"] if golden else None
    n, kinds = 0, set()
    for i, (band, n_reads, T, Lh, want_ll, use_mapq, cap) in enumerate(SCENARIOS):
        sc = scenario(rng, n_reads, T, Lh, band)
        want = stored[i] if golden else realign("ref", sc, band, want_ll, use_mapq, cap)
        got = realign(lib, sc, band, want_ll, use_mapq, cap)
        assert want["rc"] == 0 and got["rc"] == 0, (want["rc"], got["rc"])
        assert want["begin"] == got["begin"] and want["end"] == got["end"], (band, [k for k in range(n_reads) if want["begin"][k] != got["begin"][k]])
        assert want["cigar"] == got["cigar"], (band, [(a, b) for a, b in zip(want["cigar"], got["cigar"]) if a != b][:3])
        assert np.max(np.abs(np.asarray(want["loglik"]) - np.asarray(got["loglik"]))) <= tol
        if want_ll:
            assert len(set(want["loglik"])) > n_reads // 3                         # (not a vector of constants)
        kinds |= {c for cig in want["cigar"] for c in cig if c in "=XID"}
        n += n_reads
    assert kinds == set("=XID"), kinds                                            # every operation the seam can emit was compared
    # reads inside repeats longer than themselves: more tied diagonals than the ABI's 15 slots (the seam re-aligns exactly those the reference's way)
    sc = repeat_scenario(rng)
    want = stored[len(SCENARIOS)] if golden else realign("ref", sc, 16)
    got = realign(lib, sc, 16)
    assert want["rc"] == 0 and got["rc"] == 0, (want["rc"], got["rc"])
    assert want["begin"] == got["begin"] and want["end"] == got["end"], [k for k in range(len(sc["reads"])) if want["begin"][k] != got["begin"][k]]
    assert want["cigar"] == got["cigar"], [(a, b) for a, b in zip(want["cigar"], got["cigar"]) if a != b][:3]
    assert np.max(np.abs(np.asarray(want["loglik"]) - np.asarray(got["loglik"]))) <= tol
    n += len(sc["reads"])
    if golden:
        return n
    sc = scenario(rng, 9, 80, 200, 16, short=True)
    a, b = realign("ref", sc, 16), realign(lib, sc, 16)
    assert a["rc"] == b["rc"] == 1 and a["ext"] == b["ext"] and a["ext"] > 0, (a["rc"], b["rc"], a["ext"], b["ext"])
    # a haplotype the device path refuses (OCT_PHMM_EUNSUPPORTED: 65,536 bases or more with device k-mer mapping): every read takes the reference's own lines - identical to the last bit
    sc = scenario(rng, 3, 3000, 65_900, 8)
    a, b = realign("ref", sc, 8, max_ops=8192), realign(lib, sc, 8, max_ops=8192)
    assert a["rc"] == b["rc"] == 0 and len(a["cigar"][0]) > 100 and a["begin"] == b["begin"] and a["end"] == b["end"] and a["cigar"] == b["cigar"] and a["loglik"] == b["loglik"]
    return n + 3
