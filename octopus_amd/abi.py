"""ctypes mirror of include/oct_phmm.h (the C ABI) plus a numpy-backed batch container.

Host-side plumbing only: nothing here computes likelihoods. `Batch` owns the numpy arrays a call borrows
and exposes them as the oct_phmm_* structs; it is shared by the product binding (octopus_amd.engine) and
by the test-only oracle binding (oracle/__init__.py).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

OK, EINVAL, EBAND, ESHORT_HAPLOTYPE, EHIP, ENODEVICE, EUNSUPPORTED, EOVERFLOW = range(8)
LOWEST = -1.7976931348623157e308


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("max_indel_error", C.c_int32),
        ("use_int_scores", C.c_int32),
        ("use_mapping_quality", C.c_int32),
        ("mapping_quality_cap", C.c_int32),
        ("mapping_quality_cap_trigger", C.c_int32),
        ("use_flank_state", C.c_int32),
        ("nuc_prior", C.c_int32),
        ("max_mapping_positions", C.c_int32),
        ("device_id", C.c_int32),
    ]

    @staticmethod
    def default(**kw) -> "Config":
        """HaplotypeLikelihoodModel::Config{} defaults (haplotype_likelihood_model.hpp:36-44)."""
        c = Config(C.sizeof(Config), 8, 0, 1, 120, -1, 1, 2, 10, 0)
        for k, v in kw.items():
            if not hasattr(c, k):
                raise AttributeError(k)
            setattr(c, k, int(v))
        return c


class Reads(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint32),
        ("bases", C.c_void_p),
        ("qualities", C.c_void_p),
        ("offsets", C.c_void_p),
        ("mapping_quality", C.c_void_p),
        ("reverse_strand", C.c_void_p),
        ("ref_begin", C.c_void_p),
        ("n_rows", C.c_uint32),
        ("row_offsets", C.c_void_p),
    ]


class Haplotypes(C.Structure):
    _fields_ = [
        ("n_haps", C.c_uint32),
        ("bases", C.c_void_p),
        ("offsets", C.c_void_p),
        ("ref_begin", C.c_void_p),
        ("gap_open", C.c_void_p),
        ("gap_extend", C.c_void_p),
        ("snv_mask_fwd", C.c_void_p),
        ("snv_prior_fwd", C.c_void_p),
        ("snv_mask_rev", C.c_void_p),
        ("snv_prior_rev", C.c_void_p),
        ("substitution_mask", C.c_void_p),
    ]


class FlankState(C.Structure):
    _fields_ = [("lhs_flank", C.c_uint32), ("rhs_flank", C.c_uint32)]


class Regions(C.Structure):
    _fields_ = [
        ("n_regions", C.c_uint32),
        ("row_offsets", C.c_void_p),
        ("hap_offsets", C.c_void_p),
        ("has_flank", C.c_void_p),
        ("flank", C.c_void_p),
    ]


class Positions(C.Structure):
    _fields_ = [("offsets", C.c_void_p), ("positions", C.c_void_p)]


class Status(C.Structure):
    _fields_ = [
        ("code", C.c_int32),
        ("hip_error", C.c_int32),
        ("hap_index", C.c_uint32),
        ("read_index", C.c_uint32),
        ("required_extension", C.c_uint32),
        ("message", C.c_char * 128),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("n_pairs", C.c_uint64),
        ("n_candidates", C.c_uint64),
        ("n_fast_path", C.c_uint64),
        ("n_dp_score_only", C.c_uint64),
        ("n_dp_traceback", C.c_uint64),
        ("band_cells", C.c_uint64),
        ("n_dp_score_only_shared", C.c_uint64),
        ("n_dp_traceback_shared", C.c_uint64),
        ("band_cells_shared", C.c_uint64),
        ("n_pairs_shared", C.c_uint64),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _cat_bytes(seqs: Sequence[bytes]) -> np.ndarray:
    return np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if seqs else np.zeros(0, np.uint8)


def _offsets(lengths) -> np.ndarray:
    o = np.zeros(len(lengths) + 1, dtype=np.uint32)
    np.cumsum(np.asarray(lengths, dtype=np.uint64), out=o[1:], dtype=np.uint64)
    return o


@dataclass
class Batch:
    """One or more populate() calls' worth of reads and haplotypes in the flat C-ABI layout."""

    # reads
    read_bases: np.ndarray          # uint8 concat
    read_quals: np.ndarray          # uint8 concat
    read_offsets: np.ndarray        # uint32 [n+1]
    mapq: np.ndarray                # uint8 [n]
    reverse: np.ndarray             # uint8 [n]
    read_ref_begin: np.ndarray      # int64 [n]
    row_offsets: Optional[np.ndarray]  # uint32 [rows+1] or None
    # haplotypes
    hap_bases: np.ndarray
    hap_offsets: np.ndarray
    hap_ref_begin: np.ndarray
    gap_open: np.ndarray            # int8 concat
    gap_extend: np.ndarray
    snv_mask_fwd: np.ndarray        # uint8 concat
    snv_prior_fwd: np.ndarray       # int8 concat
    snv_mask_rev: np.ndarray
    snv_prior_rev: np.ndarray
    # regions / flank
    region_row_offsets: Optional[np.ndarray] = None   # uint32 [G+1]
    region_hap_offsets: Optional[np.ndarray] = None   # uint32 [G+1]
    region_has_flank: Optional[np.ndarray] = None     # uint8 [G]
    region_flank: Optional[np.ndarray] = None         # uint32 [G,2]
    flank: Optional[tuple] = None                     # single-region (lhs, rhs)
    # optional precomputed mapping positions
    pos_offsets: Optional[np.ndarray] = None          # uint64 [pairs+1]
    pos_values: Optional[np.ndarray] = None           # uint32
    _keep: list = field(default_factory=list, repr=False)

    # ---- construction helpers -----------------------------------------------------------------
    @staticmethod
    def from_lists(reads: List[dict], haps: List[dict], flank: Optional[tuple] = None,
                   templates: Optional[List[List[int]]] = None) -> "Batch":
        """reads: dicts(seq: bytes, quals: array-like, mapq, reverse, begin); haps: dicts(seq, begin, gap_open,
        gap_extend, mask_fwd, prior_fwd, mask_rev, prior_rev). templates: consecutive read index groups."""
        row_offsets = None
        if templates is not None:
            flat = [i for t in templates for i in t]
            assert flat == list(range(len(reads))), "reads of a template must be consecutive and cover all reads"
            row_offsets = _offsets([len(t) for t in templates])
        i8 = lambda key: (np.concatenate([np.asarray(h[key], dtype=np.int8) for h in haps])
                          if haps else np.zeros(0, np.int8))
        return Batch(
            read_bases=_cat_bytes([r["seq"] for r in reads]),
            read_quals=(np.concatenate([np.asarray(r["quals"], dtype=np.uint8) for r in reads])
                        if reads else np.zeros(0, np.uint8)),
            read_offsets=_offsets([len(r["seq"]) for r in reads]),
            mapq=np.asarray([r.get("mapq", 60) for r in reads], dtype=np.uint8),
            reverse=np.asarray([1 if r.get("reverse", False) else 0 for r in reads], dtype=np.uint8),
            read_ref_begin=np.asarray([r["begin"] for r in reads], dtype=np.int64),
            row_offsets=row_offsets,
            hap_bases=_cat_bytes([h["seq"] for h in haps]),
            hap_offsets=_offsets([len(h["seq"]) for h in haps]),
            hap_ref_begin=np.asarray([h.get("begin", 0) for h in haps], dtype=np.int64),
            gap_open=i8("gap_open"), gap_extend=i8("gap_extend"),
            snv_mask_fwd=_cat_bytes([bytes(h["mask_fwd"]) for h in haps]), snv_prior_fwd=i8("prior_fwd"),
            snv_mask_rev=_cat_bytes([bytes(h["mask_rev"]) for h in haps]), snv_prior_rev=i8("prior_rev"),
            flank=flank,
        )

    # ---- shape queries ------------------------------------------------------------------------
    @property
    def n_reads(self) -> int:
        return len(self.read_offsets) - 1

    @property
    def n_haps(self) -> int:
        return len(self.hap_offsets) - 1

    @property
    def n_rows(self) -> int:
        return self.n_reads if self.row_offsets is None else len(self.row_offsets) - 1

    def region_tables(self):
        if self.region_row_offsets is None:
            return np.array([0, self.n_rows], np.uint32), np.array([0, self.n_haps], np.uint32)
        return self.region_row_offsets, self.region_hap_offsets

    def hap_out_offsets(self) -> np.ndarray:
        """Offset of every haplotype's likelihood column in the flat output (column = the region's rows)."""
        ro, ho = self.region_tables()
        rows = np.repeat((ro[1:] - ro[:-1]).astype(np.uint64), (ho[1:] - ho[:-1]).astype(np.int64))
        return np.concatenate([np.zeros(1, np.uint64), np.cumsum(rows, dtype=np.uint64)])

    def out_size(self) -> int:
        ro, ho = self.region_tables()
        return int(np.sum((ro[1:] - ro[:-1]).astype(np.int64) * (ho[1:] - ho[:-1]).astype(np.int64)))

    def n_read_pairs(self) -> int:
        ro, ho = self.region_tables()
        first = (lambda row: row) if self.row_offsets is None else (lambda row: int(self.row_offsets[row]))
        return sum((first(int(ro[g + 1])) - first(int(ro[g]))) * int(ho[g + 1] - ho[g]) for g in range(len(ro) - 1))

    def read_pairs(self):
        """(haplotype, read) index pairs in the library's pair order: haplotype-major, each with the reads of its region."""
        ro, ho = self.region_tables()
        first = (lambda row: row) if self.row_offsets is None else (lambda row: int(self.row_offsets[row]))
        for g in range(len(ro) - 1):
            for h in range(int(ho[g]), int(ho[g + 1])):
                for r in range(first(int(ro[g])), first(int(ro[g + 1]))):
                    yield h, r

    # ---- ctypes views -------------------------------------------------------------------------
    def c_reads(self) -> Reads:
        for name in ("read_bases", "read_quals", "read_offsets", "mapq", "reverse", "read_ref_begin"):
            setattr(self, name, np.ascontiguousarray(getattr(self, name)))
        if self.row_offsets is not None:
            self.row_offsets = np.ascontiguousarray(self.row_offsets, dtype=np.uint32)
        return Reads(self.n_reads, _ptr(self.read_bases), _ptr(self.read_quals), _ptr(self.read_offsets),
                     _ptr(self.mapq), _ptr(self.reverse), _ptr(self.read_ref_begin), self.n_rows, _ptr(self.row_offsets))

    def without_penalty_vectors(self) -> "Batch":
        """The same batch with its six per-haplotype vectors left to the library (all six pointers NULL: oct_phmm_set_error_model)."""
        import copy
        b = copy.copy(self)
        b._keep = []
        for name in ("gap_open", "gap_extend", "snv_mask_fwd", "snv_prior_fwd", "snv_mask_rev", "snv_prior_rev"):
            setattr(b, name, None)
        return b

    def c_haps(self) -> Haplotypes:
        for name in ("hap_bases", "hap_offsets", "hap_ref_begin", "gap_open", "gap_extend", "snv_mask_fwd",
                     "snv_prior_fwd", "snv_mask_rev", "snv_prior_rev"):
            if getattr(self, name) is not None:
                setattr(self, name, np.ascontiguousarray(getattr(self, name)))
        sub = getattr(self, "substitution_mask", None)
        if sub is not None:
            self.substitution_mask = sub = np.ascontiguousarray(sub, dtype=np.uint8)
        return Haplotypes(self.n_haps, _ptr(self.hap_bases), _ptr(self.hap_offsets), _ptr(self.hap_ref_begin),
                          _ptr(self.gap_open), _ptr(self.gap_extend), _ptr(self.snv_mask_fwd), _ptr(self.snv_prior_fwd),
                          _ptr(self.snv_mask_rev), _ptr(self.snv_prior_rev), _ptr(sub))

    def c_regions(self) -> Optional[Regions]:
        if self.region_row_offsets is None:
            return None
        self.region_row_offsets = np.ascontiguousarray(self.region_row_offsets, dtype=np.uint32)
        self.region_hap_offsets = np.ascontiguousarray(self.region_hap_offsets, dtype=np.uint32)
        hf = fl = None
        if self.region_has_flank is not None:
            self.region_has_flank = np.ascontiguousarray(self.region_has_flank, dtype=np.uint8)
            self.region_flank = np.ascontiguousarray(self.region_flank, dtype=np.uint32)
            hf, fl = _ptr(self.region_has_flank), _ptr(self.region_flank)
        return Regions(len(self.region_row_offsets) - 1, _ptr(self.region_row_offsets), _ptr(self.region_hap_offsets), hf, fl)

    def c_flank(self) -> Optional[FlankState]:
        return None if self.flank is None else FlankState(int(self.flank[0]), int(self.flank[1]))

    def c_positions(self) -> Optional[Positions]:
        if self.pos_offsets is None:
            return None
        self.pos_offsets = np.ascontiguousarray(self.pos_offsets, dtype=np.uint64)
        self.pos_values = np.ascontiguousarray(self.pos_values, dtype=np.uint32)
        return Positions(_ptr(self.pos_offsets), _ptr(self.pos_values))

    def c_args(self):
        """(reads*, haps*, regions*|None, flank*|None, positions*|None) ready for the populate-style entry points."""
        r, h = self.c_reads(), self.c_haps()
        g, f, p = self.c_regions(), self.c_flank(), self.c_positions()
        self._keep = [r, h, g, f, p]
        byref = lambda s: None if s is None else C.byref(s)
        return byref(r), byref(h), byref(g), byref(f), byref(p)


class GenotypeSets(C.Structure):
    """oct_phmm_genotype_sets: genotype vectors to read out of a resident likelihood matrix."""
    _fields_ = [("n_sets", C.c_uint32), ("ploidy", C.c_void_p), ("gt_offsets", C.c_void_p), ("hap_indices", C.c_void_p),
                ("row_begin", C.c_void_p), ("row_end", C.c_void_p)]

    @staticmethod
    def make(sets):
        """sets: list of dict(genotypes = [n, ploidy] array of batch haplotype indices, rows = (begin, end) or None).
        Returns (struct, keep-alive arrays, total genotypes)."""
        ploidy = np.asarray([np.asarray(s["genotypes"]).reshape(len(s["genotypes"]), -1).shape[1] if len(s["genotypes"]) else s.get("ploidy", 1)
                             for s in sets], np.uint32)
        counts = [len(s["genotypes"]) for s in sets]
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
        idx = np.concatenate([np.asarray(s["genotypes"], np.uint32).reshape(-1) for s in sets] + [np.zeros(0, np.uint32)]).astype(np.uint32)
        keep = [ploidy, offs, idx]
        rb = re = None
        if any(s.get("rows") is not None for s in sets):
            if not all(s.get("rows") is not None for s in sets):
                raise ValueError("give rows for every set or for none")
            rb = np.asarray([s["rows"][0] for s in sets], np.uint32); re = np.asarray([s["rows"][1] for s in sets], np.uint32)
            keep += [rb, re]
        return GenotypeSets(len(sets), _ptr(ploidy), _ptr(offs), _ptr(idx), _ptr(rb), _ptr(re)), keep, int(offs[-1])


CIGAR_OPS = {1: "I", 2: "D", 7: "=", 8: "X"}


class Alignments(C.Structure):
    """oct_phmm_alignments: caller-owned outputs of oct_phmm_align."""
    _fields_ = [("max_cigar_ops", C.c_uint32), ("mapping_position", C.c_void_p), ("likelihood", C.c_void_p),
                ("n_cigar_ops", C.c_void_p), ("cigar", C.c_void_p)]

    @staticmethod
    def make(n_pairs: int, max_cigar_ops: int):
        arrays = dict(mapping_position=np.zeros(max(n_pairs, 1), np.uint32), likelihood=np.zeros(max(n_pairs, 1), np.float64),
                      n_cigar_ops=np.zeros(max(n_pairs, 1), np.uint32), cigar=np.zeros(max(n_pairs * max_cigar_ops, 1), np.uint32))
        st = Alignments(max_cigar_ops, _ptr(arrays["mapping_position"]), _ptr(arrays["likelihood"]), _ptr(arrays["n_cigar_ops"]), _ptr(arrays["cigar"]))
        return st, arrays


def alignments_result(arrays: dict, n_pairs: int, max_cigar_ops: int) -> dict:
    """Trim the output arrays and decode every CIGAR to its string form."""
    n = arrays["n_cigar_ops"][:n_pairs]
    cig = arrays["cigar"][:n_pairs * max_cigar_ops].reshape(n_pairs, max_cigar_ops) if n_pairs else np.zeros((0, max_cigar_ops), np.uint32)
    strings = ["".join(f"{int(v) >> 4}{CIGAR_OPS.get(int(v) & 15, '?')}" for v in cig[e, :min(int(n[e]), max_cigar_ops)]) for e in range(n_pairs)]
    return dict(mapping_position=arrays["mapping_position"][:n_pairs].copy(), likelihood=arrays["likelihood"][:n_pairs].copy(),
                n_cigar_ops=n.copy(), cigar=cig.copy(), cigar_strings=strings)


class ErrorModel(C.Structure):
    """oct_phmm_error_model (include/oct_phmm.h): the error models' tables as the reference's constructors expand them."""
    _fields_ = [(n, C.c_int8 * 50) for n in ("at_homopolymer_open", "cg_homopolymer_open", "dinucleotide_open", "trinucleotide_open",
                                            "homopolymer_extend", "dinucleotide_extend", "trinucleotide_extend")] + \
               [("snv_caps", (C.c_int8 * 51) * 3), ("use_snv_model", C.c_int32)]

    @staticmethod
    def make(at_open, cg_open, di_open, tri_open, snv_caps, homo_ext=(3, 3, 3, 3, 3, 3, 4, 5, 6, 6, 8, 8, 7, 6, 5, 4, 3),
             di_ext=(3, 3, 5, 4, 3, 2), tri_ext=(3, 3, 5, 4, 3, 2)):
        m = ErrorModel()
        def fill(dst, src, n):                      # copy(): first min(size, N) entries, the rest = the last one
            for i in range(n):
                dst[i] = src[i] if i < len(src) else src[-1]
        for name, src in (("at_homopolymer_open", at_open), ("cg_homopolymer_open", cg_open), ("dinucleotide_open", di_open),
                          ("trinucleotide_open", tri_open), ("homopolymer_extend", homo_ext), ("dinucleotide_extend", di_ext),
                          ("trinucleotide_extend", tri_ext)):
            fill(getattr(m, name), src, 50)
        for p in range(3):
            fill(m.snv_caps[p], snv_caps[p], 51)
        m.use_snv_model = 1
        return m
