"""CPU ORACLE bindings — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package. It binds
  * oracle/liboracle.so      — our plain-C restatement of the reference algorithm (phmm_oracle.c)
  * oracle/_ref/libref_phmm.so — the reference's own SIMD pair-HMM headers compiled in place (built in the
                                 container where /root/reference exists; travels to the GPU box as a .so)
The product package (octopus_amd) never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path
from typing import Optional

import numpy as np

from octopus_amd import abi

_DIR = Path(__file__).resolve().parent
_LIB: Optional[C.CDLL] = None
_REF: Optional[C.CDLL] = None
ISA = {"sse2": 0, "avx2": 1, "avx512": 2, "native": -1}


def build(quiet: bool = True) -> None:
    """make liboracle.so (always), _ref/libref_phmm.so + libref_array*.so and the INTEGRATION-patched seams against whichever builds of
    the C ABI exist (only where /root/reference is present). Raises when make fails OR when a library this invocation is responsible for
    (`make expected`) is not there afterwards: a build that reports success without its libraries would turn parity tests into skips."""
    make = ["make", "-C", str(_DIR), "--no-print-directory"]
    subprocess.run(make + ["-j4", "all", "patched"], check=True, stdout=subprocess.DEVNULL if quiet else None)
    expected = subprocess.run(make + ["expected"], check=True, capture_output=True, text=True).stdout.split()
    missing = [name for name in expected if not (_DIR / name).exists()]
    if missing:
        raise RuntimeError(f"oracle.build(): make succeeded but {missing} are missing")


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        p = _DIR / "liboracle.so"
        if not p.exists():
            build()
        _LIB = C.CDLL(str(p))
        _LIB.oracle_align.restype = C.c_int
        _LIB.oracle_flank.restype = C.c_int
        _LIB.oracle_evaluate.restype = C.c_double
        _LIB.oracle_time_align_windows.restype = C.c_double
    return _LIB


def have_ref() -> bool:
    return (_DIR / "_ref" / "libref_phmm.so").exists()


def ref() -> C.CDLL:
    global _REF
    if _REF is None:
        _REF = C.CDLL(str(_DIR / "_ref" / "libref_phmm.so"))
        _REF.ref_phmm_align.restype = C.c_int
        _REF.ref_phmm_flank.restype = C.c_int
    return _REF


_REF_ARRAYS: dict = {}


def have_ref_array(isa: str = "sse2") -> bool:
    """oracle/_ref/libref_array.so (SSE2 kernels) or libref_array_avx2.so (the reference's AVX2 kernels; needs an AVX2 CPU)."""
    name = "libref_array.so" if isa == "sse2" else "libref_array_avx2.so"
    return (_DIR / "_ref" / name).exists() and (isa == "sse2" or ref_isa_supported("avx2"))


_ARRAY_LIBS = {"sse2": "libref_array.so", "avx2": "libref_array_avx2.so",
               # the reference's class with INTEGRATION.md's patch applied (oracle/make_patched_tree.py), on the simulator / GPU build of the C ABI
               "patched_sim": "libref_array_patched_sim.so", "patched_gpu": "libref_array_patched_gpu.so"}


def have_patched_array(backend: str) -> bool:
    return (_DIR / "_ref" / _ARRAY_LIBS["patched_" + backend]).exists()


def _ref_array_lib(isa: str = "sse2") -> C.CDLL:
    if isa not in _REF_ARRAYS:
        lib_ = C.CDLL(str(_DIR / "_ref" / _ARRAY_LIBS[isa]))
        lib_.ref_array_time_populate.restype = C.c_double
        _REF_ARRAYS[isa] = lib_
    return _REF_ARRAYS[isa]


class _RefArrayArgs(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("max_indel_error", "use_int_scores", "use_mapping_quality", "mapping_quality_cap",
                                         "mapping_quality_cap_trigger", "use_flank_state")] + [
        ("n_haps", C.c_uint32), ("hap_bases", C.c_void_p), ("hap_off", C.c_void_p), ("hap_begin", C.c_void_p),
        ("gap_open", C.c_void_p), ("gap_extend", C.c_void_p), ("mask_f", C.c_void_p), ("prior_f", C.c_void_p), ("mask_r", C.c_void_p), ("prior_r", C.c_void_p),
        ("has_flank", C.c_int32), ("lhs_flank", C.c_uint32), ("rhs_flank", C.c_uint32),
        ("n_reads", C.c_uint32), ("read_bases", C.c_void_p), ("quals", C.c_void_p), ("read_off", C.c_void_p), ("read_begin", C.c_void_p),
        ("mapq", C.c_void_p), ("reverse", C.c_void_p),
        ("n_rows", C.c_uint32), ("row_off", C.c_void_p), ("n_samples", C.c_uint32), ("sample_row_off", C.c_void_p), ("n_threads", C.c_int32)]


def _ref_array_args(cfg, batch, sample_rows, n_threads):
    assert batch.region_row_offsets is None or len(batch.region_row_offsets) == 2, "one populate() call = one region"
    n_reads = len(batch.read_offsets) - 1
    n_rows = n_reads if batch.row_offsets is None else len(batch.row_offsets) - 1
    n_haps = len(batch.hap_offsets) - 1
    srows = np.asarray([0, n_rows] if sample_rows is None else sample_rows, dtype=np.uint32)
    flank = batch.flank if batch.region_flank is None else (tuple(int(x) for x in batch.region_flank[0]) if batch.region_has_flank[0] else None)
    keep = [np.ascontiguousarray(x) for x in (batch.hap_bases, batch.hap_offsets.astype(np.uint32), batch.hap_ref_begin.astype(np.int64),
                                              batch.gap_open, batch.gap_extend, batch.snv_mask_fwd, batch.snv_prior_fwd, batch.snv_mask_rev,
                                              batch.snv_prior_rev, batch.read_bases, batch.read_quals, batch.read_offsets.astype(np.uint32),
                                              batch.read_ref_begin.astype(np.int64), batch.mapq, batch.reverse, srows)]
    rows = None if batch.row_offsets is None else np.ascontiguousarray(batch.row_offsets.astype(np.uint32))
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    a = _RefArrayArgs(cfg.max_indel_error, cfg.use_int_scores, cfg.use_mapping_quality, cfg.mapping_quality_cap, cfg.mapping_quality_cap_trigger,
                      cfg.use_flank_state, n_haps, vp(keep[0]), vp(keep[1]), vp(keep[2]), vp(keep[3]), vp(keep[4]), vp(keep[5]), vp(keep[6]),
                      vp(keep[7]), vp(keep[8]), 1 if flank is not None else 0, int(flank[0]) if flank else 0, int(flank[1]) if flank else 0,
                      n_reads, vp(keep[9]), vp(keep[10]), vp(keep[11]), vp(keep[12]), vp(keep[13]), vp(keep[14]),
                      n_rows, vp(rows) if rows is not None else None, len(srows) - 1, vp(keep[15]), int(n_threads))
    return a, (keep, rows, n_haps, n_rows)


def ref_array_time_populate(cfg: abi.Config, batch: abi.Batch, n_threads: int, reps: int = 1, isa: str = "sse2") -> float:
    """Seconds the REFERENCE's own HaplotypeLikelihoodArray::populate (TemplateMap overload, one read per template, its ThreadPool of
    n_threads workers fanning out over haplotypes, array.cpp:167-184) spends on `reps` calls of this single-region batch."""
    a, keep = _ref_array_args(cfg, batch, None, n_threads)
    return float(_ref_array_lib(isa).ref_array_time_populate(C.byref(a), int(reps)))


def ref_array_populate(cfg: abi.Config, batch: abi.Batch, sample_rows=None, n_threads: int = 1, merged: bool = False, isa: str = "sse2"):
    """The REFERENCE's own HaplotypeLikelihoodArray::populate (haplotype_likelihood_array.cpp, built in place into
    oracle/_ref/libref_array.so) on a single-region batch whose rows are split into samples at `sample_rows` (row offsets).
    Mapping positions come from the reference's own k-mer mapper (the batch's are ignored), as in the reference.
    Returns (code, out [H x rows] flat, merged or None, err_hap, required_extension): code 0 ok, 1 ShortHaplotypeError."""
    a, (keep, rows, n_haps, n_rows) = _ref_array_args(cfg, batch, sample_rows, n_threads)
    out = np.full(max(n_haps * n_rows, 1), np.nan)
    mg = np.full(max(n_haps * n_rows, 1), np.nan) if merged else None
    err_hap, ext = C.c_uint32(0), C.c_uint32(0)
    code = _ref_array_lib(isa).ref_array_populate(C.byref(a), _p(out), _p(mg) if merged else None, C.byref(err_hap), C.byref(ext))
    return code, out[:n_haps * n_rows], (mg[:n_haps * n_rows] if merged else None), err_hap.value, ext.value


def ref_array_exercise(cfg: abi.Config, batch: abi.Batch, sample_rows, keep, n_threads: int = 1, lib: str = "sse2"):
    """populate() + every read-back method of HaplotypeLikelihoodArray (oracle/ref_array_bridge.cpp: ref_array_exercise) on the unpatched
    reference class (lib "sse2"/"avx2") or on the class with INTEGRATION.md's patch applied (lib "patched_sim"/"patched_gpu").
    Returns (code, sections [6, H, rows], flags [5], err_hap, required_extension)."""
    a, (keep_alive, rows, n_haps, n_rows) = _ref_array_args(cfg, batch, sample_rows, n_threads)
    sec = np.full(6 * max(n_haps * n_rows, 1), np.nan)
    flags = np.zeros(5, np.uint32)
    kp = np.ascontiguousarray(keep, dtype=np.uint32)
    err_hap, ext = C.c_uint32(0), C.c_uint32(0)
    code = _ref_array_lib(lib).ref_array_exercise(C.byref(a), _p(kp), len(kp), _p(sec), _p(flags), C.byref(err_hap), C.byref(ext))
    return code, sec[:6 * n_haps * n_rows].reshape(6, n_haps, n_rows), flags, err_hap.value, ext.value


def ref_isa_supported(isa: str) -> bool:
    return have_ref() and bool(ref().ref_phmm_isa_supported(ISA[isa])) if isa != "native" else have_ref()


def _p(a):
    return None if a is None else np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)


def _i8(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.int8))


def _bytes(a):
    if a is None:
        return None
    if isinstance(a, (bytes, bytearray)):
        return np.frombuffer(bytes(a), dtype=np.uint8)
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint8))


def align(band, score_bits, truth, target, quals, gap_open, gap_extend=None, gap_extend_scalar=1,
          snv_mask=None, snv_prior=None, nuc_prior=2, traceback=False, backend: str = "oracle"):
    """simd::PairHMM::align. backend: 'oracle' (restatement) or 'sse2'/'avx2'/'avx512'/'native' (reference .so).
    Returns dict(score, status[, first_pos, align1, align2])."""
    truth_a, target_a = _bytes(truth), _bytes(target)
    q, go, ge = _i8(np.asarray(quals, dtype=np.uint8).view(np.int8)), _i8(gap_open), _i8(gap_extend)
    m, pr = _bytes(snv_mask), _i8(snv_prior)
    T = len(target_a)
    n = 2 * (T + band) + 1
    a1, a2 = C.create_string_buffer(n + 1), C.create_string_buffer(n + 1)
    fp, st = C.c_int(0), C.c_int(0)
    args = (band, score_bits, _p(truth_a), _p(target_a), _p(q), len(truth_a), T, _p(m), _p(pr), _p(go), _p(ge),
            int(gap_extend_scalar), int(nuc_prior), 1 if traceback else 0, C.byref(fp), a1, a2, C.byref(st))
    if backend == "oracle":
        score = lib().oracle_align(*args)
    else:
        score = ref().ref_phmm_align(ISA[backend], *args)
    out = dict(score=int(score), status=st.value)
    if traceback:
        out.update(first_pos=fp.value, align1=a1.value.decode("latin1"), align2=a2.value.decode("latin1"))
    return out


def flank(band, score_bits, truth_len, lhs, rhs, target, quals, snv_mask, snv_prior, gap_open, gap_extend, nuc_prior,
          first_pos, aln1: str, aln2: str, backend: str = "oracle"):
    """simd::PairHMM::calculate_flank_score (masked overload). Returns (flank_score, target_mask_size, status)."""
    target_a = _bytes(target)
    q = _i8(np.asarray(quals, dtype=np.uint8).view(np.int8))
    m, pr, go, ge = _bytes(snv_mask), _i8(snv_prior), _i8(gap_open), _i8(gap_extend)
    ms, st = C.c_int(0), C.c_int(0)
    args = (band, score_bits, truth_len, lhs, rhs, _p(target_a), _p(q), _p(m), _p(pr), _p(go), _p(ge), int(nuc_prior),
            int(first_pos), aln1.encode("latin1"), aln2.encode("latin1"), C.byref(ms), C.byref(st))
    s = lib().oracle_flank(*args) if backend == "oracle" else ref().ref_phmm_flank(ISA[backend], *args)
    return int(s), ms.value, st.value


def set_l1_backend(name: str = "oracle") -> None:
    """Choose the L1 kernels the oracle's L2/L3 layers drive: the restatement or the reference's SIMD kernels."""
    L = lib()
    L.oracle_set_l1_backend.argtypes = [C.c_void_p, C.c_void_p]
    if name == "oracle":
        L.oracle_set_l1_backend(None, None)
        return
    r = ref()
    a = C.cast(getattr(r, f"ref_phmm_align_bound_{name}"), C.c_void_p)
    f = C.cast(getattr(r, f"ref_phmm_flank_bound_{name}"), C.c_void_p)
    L.oracle_set_l1_backend(a, f)


def try_naive_evaluate(truth, target, quals, offset, gap_open, gap_extend, mask, prior, lhs=0, rhs=0):
    truth_a, target_a = _bytes(truth), _bytes(target)
    pen = C.c_int(0)
    handled = lib().oracle_try_naive_evaluate(_p(truth_a), len(truth_a), _p(target_a), len(target_a),
                                              _p(np.asarray(quals, np.uint8)), C.c_uint32(offset), _p(_i8(gap_open)),
                                              _p(_i8(gap_extend)), _p(_bytes(mask)), _p(_i8(prior)),
                                              C.c_uint32(lhs), C.c_uint32(rhs), C.byref(pen))
    return bool(handled), pen.value


def evaluate(truth, target, quals, offset, band, score_bits, gap_open, gap_extend, mask, prior, lhs=0, rhs=0, nuc_prior=2):
    truth_a, target_a = _bytes(truth), _bytes(target)
    kind = C.c_int(0)
    v = lib().oracle_evaluate(_p(truth_a), len(truth_a), _p(target_a), len(target_a), _p(np.asarray(quals, np.uint8)),
                              C.c_uint32(offset), band, score_bits, _p(_i8(gap_open)), _p(_i8(gap_extend)),
                              _p(_bytes(mask)), _p(_i8(prior)), C.c_uint32(lhs), C.c_uint32(rhs), int(nuc_prior), C.byref(kind))
    return float(v), kind.value


def map_query_to_target(query, target, max_positions=10):
    q, t = _bytes(query), _bytes(target)
    out = np.zeros(max(max_positions, 1), dtype=np.uint32)
    n = lib().oracle_map_query_to_target(_p(q), len(q), _p(t), len(t), int(max_positions), _p(out))
    return out[:n].tolist()


def populate(cfg: abi.Config, batch: abi.Batch, n_threads: int = 1):
    """HaplotypeLikelihoodArray::populate on the CPU oracle. Returns (out ndarray, status, stats dict)."""
    out = np.full(max(batch.out_size(), 1), np.nan, dtype=np.float64)
    st, stats = abi.Status(), abi.Stats()
    r, h, g, f, p = batch.c_args()
    code = lib().oracle_populate(C.byref(cfg), r, h, g, f, p, _p(out), C.byref(st), C.byref(stats), int(n_threads))
    assert code == st.code
    return out[:batch.out_size()], st, stats.as_dict()


def align_batch(cfg: abi.Config, batch: abi.Batch, max_cigar_ops: int = 64, n_threads: int = 1):
    """HaplotypeLikelihoodModel::align for every (haplotype, read) pair on the CPU oracle. Returns (result dict, status)."""
    n = batch.n_read_pairs()
    out, arrays = abi.Alignments.make(n, max_cigar_ops)
    st = abi.Status()
    r, h, g, f, p = batch.c_args()
    code = lib().oracle_align_batch(C.byref(cfg), r, h, g, f, p, C.byref(out), C.byref(st), int(n_threads))
    assert code == st.code
    return abi.alignments_result(arrays, n, max_cigar_ops), st


def genotype_likelihoods(lik, hap_out_off, genotypes, rows=None):
    """ConstantMixtureGenotypeLikelihoodModel::evaluate for an [n, ploidy] array of sorted haplotype indices over the
    flat matrix `lik` (populate's output); rows = (begin, end) within the region, default all."""
    g = np.ascontiguousarray(np.asarray(genotypes, np.uint32).reshape(len(genotypes), -1))
    off = np.ascontiguousarray(hap_out_off, dtype=np.uint64)
    if rows is None:
        rows = (0, int(off[int(g[0, 0]) + 1] - off[int(g[0, 0])])) if len(g) else (0, 0)
    out = np.empty(max(len(g), 1), np.float64)
    code = lib().oracle_genotype_likelihoods(_p(np.ascontiguousarray(lik, dtype=np.float64)), _p(off), len(g), g.shape[1] if len(g) else 1,
                                             _p(g), int(rows[0]), int(rows[1]), _p(out))
    assert code == 0, code
    return out[:len(g)]


ErrorModel = abi.ErrorModel      # oct_phmm_error_model is part of the C ABI (include/oct_phmm.h)


def tandem_repeats(seq, min_period=1, max_period=5, backend="oracle"):
    """tandem::extract_exact_tandem_repeats -> [(pos, length, period)] from the restatement or the reference's library (oracle/_ref)."""
    s = bytes(seq)
    cap = 4 * len(s) + 64
    out = (C.c_uint32 * (3 * cap))()
    n = (lib().oracle_tandem_repeats if backend == "oracle" else ref().ref_tandem_repeats)(s, len(s), int(min_period), int(max_period), out, cap)
    assert n <= cap
    return [(out[3 * i], out[3 * i + 1], out[3 * i + 2]) for i in range(n)]


def penalty_vectors(model: "ErrorModel", seq, substitution_mask=None):
    """HaplotypeLikelihoodModel::reset's six vectors for one haplotype sequence: (gap_open, gap_extend, mask_fwd, prior_fwd, mask_rev, prior_rev)."""
    s = bytes(seq); n = len(s)
    go, ge, pf, pr = (np.zeros(n, np.int8) for _ in range(4))
    mf, mr = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    lib().oracle_indel_penalties(C.byref(model), s, n, _p(go), _p(ge))
    lib().oracle_snv_priors(C.byref(model), s, n, _p(None if substitution_mask is None else np.asarray(substitution_mask, np.uint8)),
                            _p(mf), _p(pf), _p(mr), _p(pr))
    return go, ge, mf, pf, mr, pr


def time_align_windows(band, score_bits, truth, truth_offsets, target, quals, target_offsets, gap_open, gap_extend,
                       snv_mask, snv_prior, nuc_prior=2, traceback=False, reps=1, n_threads=1):
    """Seconds to run the current L1 backend over all windows `reps` times (CPU baseline leg of bench.py)."""
    chk = C.c_int64(0)
    n = len(truth_offsets) - 1
    secs = lib().oracle_time_align_windows(band, score_bits, C.c_uint32(n), _p(_bytes(truth)),
                                           _p(np.asarray(truth_offsets, np.uint32)), _p(_bytes(target)),
                                           _p(np.asarray(quals, np.uint8)), _p(np.asarray(target_offsets, np.uint32)),
                                           _p(_i8(gap_open)), _p(_i8(gap_extend)), _p(_bytes(snv_mask)), _p(_i8(snv_prior)),
                                           int(nuc_prior), 1 if traceback else 0, int(reps), int(n_threads), C.byref(chk))
    return float(secs), int(chk.value)


def host_cores() -> int:
    """Host threads the CPU baseline may use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes expose 256 logical
    CPUs but grant a 16-CPU quota; oversubscribing it only adds throttling)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n
