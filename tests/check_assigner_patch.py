"""INTEGRATION.md's second seam, compiled: the reference's OWN read-assignment likelihood functions (src/core/tools/read_assigner.cpp:145-287 - expand every
haplotype of the genotype, k-mer table, model.reset, model.evaluate read by read; cut out of a copy of the file by oracle/make_patched_tree.py and
compiled between stand-in types in oracle/ref_assigner_bridge.cpp) against the same functions with the last one replaced by
integration/read_assigner_on_device.inc (expand -> reset -> pack -> ONE oct_phmm_populate with the device's k-mer mapper). Both sides run the
reference's real repeat-based error models on the expanded haplotypes; reads and templates, ploidies 2 - 4, haplotypes with indels against the reference
(so that the expansion's indel factor is not zero), and reads so far outside the haplotypes that both give up the same way. The reference runs its SERIAL branch
here: its thread-pool branch (:260-270) hands ONE mutable lambda - and with it one HaplotypeLikelihoodModel - to all pool threads (utils/parallel_transform.hpp:116-118
captures `op` by reference), which reset() and evaluate() it concurrently: a data race that crashes or changes results under load (reproduced here with 8 threads)."""
import ctypes as C
import json
from pathlib import Path

import numpy as np

import oracle

ROOT = Path(__file__).resolve().parents[1]
_LIBS = {"ref": "libref_assigner.so", "patched_sim": "libref_assigner_patched_sim.so", "patched_gpu": "libref_assigner_patched_gpu.so"}
_loaded = {}


def have(which: str) -> bool:
    return (ROOT / "oracle" / "_ref" / _LIBS[which]).exists()


def _lib(which):
    if which not in _loaded:
        _loaded[which] = C.CDLL(str(ROOT / "oracle" / "_ref" / _LIBS[which]))
    return _loaded[which]


class _Args(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("max_indel_error", "use_int_scores", "use_mapping_quality", "mapping_quality_cap", "mapping_quality_cap_trigger")] + [
        ("tables", C.c_void_p), ("table_lens", C.c_void_p),
        ("ploidy", C.c_uint32), ("hap_bases", C.c_void_p), ("hap_off", C.c_void_p), ("hap_begin", C.c_void_p), ("hap_region_size", C.c_void_p),
        ("left_bases", C.c_void_p), ("left_off", C.c_void_p), ("right_bases", C.c_void_p), ("right_off", C.c_void_p),
        ("n_reads", C.c_uint32), ("read_bases", C.c_void_p), ("quals", C.c_void_p), ("read_off", C.c_void_p), ("read_begin", C.c_void_p),
        ("mapq", C.c_void_p), ("reverse", C.c_void_p), ("n_rows", C.c_uint32), ("row_off", C.c_void_p), ("n_threads", C.c_int32)]


def default_tables():
    t = json.loads((ROOT / "tests" / "golden" / "error_model_tables.json").read_text())
    tabs = t["indel_open"][t["default"]] + t["extend"] + t["snv_caps"][t["default"].split(".")[0]]
    return np.asarray([v for tab in tabs for v in tab], np.int8), np.asarray([len(tab) for tab in tabs], np.uint32)


def scenario(rng, ploidy, n_reads, T, span, templates, context=260):
    """A stretch of reference, `ploidy` haplotypes over [start, start + span) with a few substitutions and one indel each, reads drawn from them."""
    acgt = np.frombuffer(b"ACGT", np.uint8)
    ref = acgt[rng.integers(0, 4, span + 2 * context + 40)]
    ref[context + 30:context + 42] = ord("A")                                  # a homopolymer and a dinucleotide run: the error models have something to price
    ref[context + 90:context + 106] = np.frombuffer(b"CA" * 8, np.uint8)
    start = 10_000
    haps, lefts, rights, begins, spans = [], [], [], [], []
    for k in range(ploidy):
        seq = ref[context:context + span].copy()
        for _ in range(3):
            seq[int(rng.integers(5, span - 5))] = acgt[int(rng.integers(0, 4))]
        p = int(rng.integers(20, span - 20)); n = int(rng.integers(1, 5))
        seq = np.concatenate([seq[:p], acgt[rng.integers(0, 4, n)], seq[p:]]) if k % 2 == 0 else np.concatenate([seq[:p], seq[p + n:]])
        haps.append(seq); lefts.append(ref[:context]); rights.append(ref[context + span:]); begins.append(start); spans.append(span)
    reads, quals, rbegin = [], [], []
    for r in range(n_reads):
        h = haps[int(rng.integers(0, ploidy))]
        o = int(rng.integers(0, len(h) - T))
        s = h[o:o + T].copy()
        q = rng.choice(np.asarray([12, 25, 37], np.uint8), size=T, p=[0.1, 0.2, 0.7])
        flip = rng.random(T) < np.where(q == 37, 0.002, np.where(q == 25, 0.01, 0.08))
        s[flip] = acgt[rng.integers(0, 4, int(flip.sum()))]
        reads.append(s); quals.append(q); rbegin.append(start + o + int(rng.integers(-2, 3)))
    row_off = None
    if templates:
        rows, r = [0], 0
        while r < n_reads:
            r += min(n_reads - r, int(rng.integers(1, 3)))
            rows.append(r)
        row_off = np.asarray(rows, np.uint32)
    return dict(haps=haps, lefts=lefts, rights=rights, begins=np.asarray(begins, np.int64), spans=np.asarray(spans, np.uint32),
                reads=reads, quals=quals, rbegin=np.asarray(rbegin, np.int64), mapq=rng.integers(0, 70, n_reads).astype(np.uint8),
                reverse=rng.integers(0, 2, n_reads).astype(np.uint8), row_off=row_off)


def likelihoods(which, sc, band, n_threads=1, mapq_cap=None):
    flat, lens = default_tables()
    cat = lambda xs: (np.concatenate(xs).astype(np.uint8), np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.uint32))
    hb, ho = cat(sc["haps"]); lb, lo = cat(sc["lefts"]); rb, ro = cat(sc["rights"]); qb, qo = cat(sc["reads"]); ql, _ = cat(sc["quals"])
    n_reads = len(sc["reads"]); n_rows = n_reads if sc["row_off"] is None else len(sc["row_off"]) - 1
    keep = [flat, lens, hb, ho, lb, lo, rb, ro, qb, qo, ql]
    p = lambda a: None if a is None else np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    a = _Args(band, 0, 1, 40 if mapq_cap else 255, 30 if mapq_cap else -1, p(flat), p(lens),
              len(sc["haps"]), p(hb), p(ho), p(sc["begins"]), p(sc["spans"]), p(lb), p(lo), p(rb), p(ro),
              n_reads, p(qb), p(ql), p(qo), p(sc["rbegin"]), p(sc["mapq"]), p(sc["reverse"]), n_rows, p(sc["row_off"]), n_threads)
    out = np.full((len(sc["haps"]), n_rows), np.nan)
    ext = C.c_uint32(0)
    rc = _lib(which).ref_assigner_likelihoods(C.byref(a), p(out), C.byref(ext))
    del keep
    return rc, out, ext.value


GOLDEN = ROOT / "tests" / "golden" / "assigner_seam_golden.json"
SCENARIOS = ((8, 2, 24, 60, 170, False, 1, False), (16, 3, 30, 100, 240, True, 1, True), (16, 2, 20, 150, 300, False, 1, False), (32, 4, 16, 120, 330, True, 1, False))   # (threads: 1 = the reference's serial branch, see above)


def check(backend, tol=0.0, golden=False):
    """golden: compare with the matrices the reference's functions produced where tests/golden/make_assigner_seam_golden.py ran (committed), instead of calling
    them here - the GPU box's run, which needs only the patched library."""
    lib = "patched_" + backend
    rng = np.random.default_rng(77)
    stored = json.loads(GOLDEN.read_text())["matrices"] if golden else None
    n = 0
    for i, (band, ploidy, n_reads, T, span, templates, threads, cap) in enumerate(SCENARIOS):
        sc = scenario(rng, ploidy, n_reads, T, span, templates)
        rc0, want = (0, np.asarray(stored[i])) if golden else likelihoods("ref", sc, band, threads, cap)[:2]
        rc1, got, _ = likelihoods(lib, sc, band, 1, cap)
        assert rc0 == 0 and rc1 == 0, (rc0, rc1)
        assert want.shape == got.shape
        assert not np.isnan(want).any() and np.max(np.abs(want - got)) <= tol, (band, templates, np.max(np.abs(want - got)))
        assert np.unique(want).size > want.size // 4                              # (not a matrix of constants)
        n += want.size
    if golden:
        return n
    # a read far outside everything expand() can reach: both sides fail, the same way (the seam's own expansion always covers its reads, so ShortHaplotypeError is
    # out of reach here; the populate patch's test covers that exception)
    sc = scenario(rng, 2, 8, 60, 150, False, context=40)
    sc["rbegin"][3] = 10_000 - 90
    a = likelihoods("ref", sc, 16)
    b = likelihoods(lib, sc, 16)
    assert a[0] == b[0] and a[0] in (1, 2) and (a[0] != 1 or a[2] == b[2]), (a[0], b[0], a[2], b[2])
    # haplotypes the device path refuses (OCT_PHMM_EUNSUPPORTED: 65,536 bases or more with device k-mer mapping): the reference's own function, kept as calculate_likelihoods_on_host, answers - identical to the last bit
    sc = scenario(rng, 2, 4, 400, 65_900, False)
    a = likelihoods("ref", sc, 8)
    b = likelihoods(lib, sc, 8)
    assert a[0] == b[0] == 0 and np.array_equal(a[1], b[1]) and not np.isnan(a[1]).any(), (a[0], b[0])
    return n + a[1].size
