"""ctypes binding of liboct_phmm.so (include/oct_phmm.h) used by tests/ and bench.py.

Plumbing only. `load()` fails loudly when the HIP library has not been built; there is no CPU path here.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path
from typing import Optional

import numpy as np

from . import abi

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "liboct_phmm.so"
HIPCC = "/opt/rocm/bin/hipcc"


class EngineError(RuntimeError):
    def __init__(self, code: int, status: Optional[abi.Status] = None, what: str = ""):
        self.code = code
        self.status = status
        msg = status.message.decode("latin1") if status is not None else ""
        super().__init__(f"oct_phmm error {code} {what}: {msg}")


def source_digest(files, flags) -> str:
    """sha256 over the contents of `files` and the compiler flags: what a built artefact is stamped with (a sidecar `<artefact>.srchash`), so that a
    stale binary is never reused because its mtime happens to be newer than the sources' (a checkout, a copied tree, a clock step)."""
    import hashlib
    m = hashlib.sha256()
    for f in files:
        m.update(Path(f).name.encode()); m.update(Path(f).read_bytes())
    m.update(" ".join(flags).encode())
    return m.hexdigest()


def up_to_date(artefact: Path, digest: str) -> bool:
    stamp = Path(str(artefact) + ".srchash")
    return artefact.exists() and stamp.exists() and stamp.read_text().strip() == digest


def stamp(artefact: Path, digest: str) -> None:
    Path(str(artefact) + ".srchash").write_text(digest + "\n")


def build(force: bool = False) -> Path:
    """hipcc --offload-arch=gfx950 the kernels + host API into octopus_amd/liboct_phmm.so (in-tree). Rebuilt whenever the sources' digest differs from
    the one the library was stamped with."""
    srcs = [PKG_DIR / "csrc" / "oct_phmm.hip"] + sorted((PKG_DIR / "csrc").glob("*.hpp")) + sorted((PKG_DIR / "csrc").glob("*.hh")) + sorted((PKG_DIR / "csrc").glob("*.h"))
    # (every file the one translation unit includes: *.hpp = device code and its data model, host_*.hh = the host side by subsystem, *.h = the model-file reader)
    srcs.append(PKG_DIR.parent / "include" / "oct_phmm.h")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]
    digest = source_digest(srcs, flags)
    if not force and up_to_date(LIB_PATH, digest):
        return LIB_PATH
    subprocess.run([HIPCC] + flags + [str(srcs[0]), "-o", str(LIB_PATH)], check=True)
    stamp(LIB_PATH, digest)
    return LIB_PATH


def kernel_source_sha() -> str:
    """sha256 over the kernel sources of liboct_phmm.so (octopus_amd/csrc/*.hpp): stamps profile summaries (tools/summarize_pmc.py) so that bench.py can tell
    whether committed counter values belong to the kernels it is timing (the GPU box has no .git to ask)."""
    import hashlib
    m = hashlib.sha256()
    for f in sorted((PKG_DIR / "csrc").glob("*.hpp")):          # the device code: kernels, intrinsics layer, device data model (not the host API file)
        m.update(f.name.encode()); m.update(f.read_bytes())
    return m.hexdigest()[:16]


_LIBS = {}


def load(path: Optional[Path] = None) -> C.CDLL:
    import os
    p = Path(path) if path is not None else Path(os.environ.get("OCT_PHMM_LIB", LIB_PATH))      # OCT_PHMM_LIB: A/B of two builds in one GPU session (tools/)
    if not p.exists():
        raise FileNotFoundError(f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    key = str(p)
    if key not in _LIBS:
        lib = C.CDLL(key)
        lib.oct_phmm_create.argtypes = [C.POINTER(abi.Config), C.POINTER(C.c_void_p)]
        lib.oct_phmm_destroy.argtypes = [C.c_void_p]
        lib.oct_phmm_band_size.argtypes = [C.c_void_p]
        lib.oct_phmm_strerror.restype = C.c_char_p
        pv = C.c_void_p
        lib.oct_phmm_populate.argtypes = [pv, pv, pv, pv, pv, pv, pv, pv]
        lib.oct_phmm_batch_upload.argtypes = [pv, pv, pv, pv, pv, pv, C.POINTER(C.c_void_p), pv]
        lib.oct_phmm_batch_run.argtypes = [pv, pv, pv]
        lib.oct_phmm_batch_wait.argtypes = [pv, pv, pv]
        lib.oct_phmm_batch_download.argtypes = [pv, pv, pv, pv]
        lib.oct_phmm_batch_stats.argtypes = [pv, pv]
        lib.oct_phmm_batch_candidate_positions.argtypes = [pv, pv, pv, pv, pv]
        lib.oct_phmm_batch_out_size.argtypes = [pv]
        lib.oct_phmm_batch_out_size.restype = C.c_size_t
        lib.oct_phmm_batch_device_sized.argtypes = [pv]
        lib.oct_phmm_probe_clock.argtypes = [pv, C.c_double, C.POINTER(C.c_double)]
        lib.oct_phmm_test_set.argtypes = [C.c_char_p, C.c_char_p]
        lib.oct_phmm_batch_kernel_time.argtypes = [pv, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
        lib.oct_phmm_batch_kernel_time_by_kind.argtypes = [pv, C.POINTER(C.c_double * 4), C.POINTER(C.c_uint32 * 4)]
        lib.oct_phmm_batch_genotype_likelihoods.argtypes = [pv, pv, pv, pv, pv]
        lib.oct_phmm_align.argtypes = [pv, pv, pv, pv, pv, pv, pv, pv]
        lib.oct_phmm_align_candidate_counts.argtypes = [pv, pv, C.c_size_t, C.POINTER(C.c_uint32)]
        lib.oct_phmm_set_timing.argtypes = [pv, C.c_int]
        lib.oct_phmm_server_create.argtypes = [C.POINTER(abi.Config), C.c_uint32, C.POINTER(C.c_void_p)]
        lib.oct_phmm_server_destroy.argtypes = [pv]
        lib.oct_phmm_server_create_multi.argtypes = [C.POINTER(abi.Config), C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        lib.oct_phmm_server_device_calls.argtypes = [pv, C.POINTER(C.c_uint64), C.c_uint32]
        lib.oct_phmm_server_populate.argtypes = [pv, pv, pv, pv, pv, pv, pv]
        lib.oct_phmm_server_stats.argtypes = [pv, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        lib.oct_phmm_batch_free.argtypes = [pv, pv]
        lib.oct_phmm_error_model_default.argtypes = [C.POINTER(abi.ErrorModel)]
        lib.oct_phmm_penalty_vectors.argtypes = [C.POINTER(abi.ErrorModel), C.c_uint32] + [pv] * 10
        lib.oct_phmm_set_error_model.argtypes = [pv, C.POINTER(abi.ErrorModel)]
        lib.oct_phmm_batch_penalty_vectors.argtypes = [pv] * 9
        lib.oct_phmm_server_set_error_model.argtypes = [pv, C.POINTER(abi.ErrorModel)]
        lib.oct_phmm_custom_indel_model_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
        lib.oct_phmm_custom_indel_model_create.argtypes = [pv, C.c_uint32, C.c_int8, pv, C.c_uint32, C.c_int32, C.c_int8, C.POINTER(C.c_void_p)]
        lib.oct_phmm_custom_indel_model_destroy.argtypes = [pv]; lib.oct_phmm_custom_indel_model_destroy.restype = None
        lib.oct_phmm_custom_indel_model_info.argtypes = [pv, C.POINTER(C.c_int8), C.POINTER(C.c_int8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
        lib.oct_phmm_custom_penalty_vectors.argtypes = [pv, C.POINTER(abi.ErrorModel), C.c_uint32] + [pv] * 10
        lib.oct_phmm_set_custom_error_model.argtypes = [pv, pv, C.POINTER(abi.ErrorModel)]
        lib.oct_phmm_server_set_custom_error_model.argtypes = [pv, pv, C.POINTER(abi.ErrorModel)]
        _LIBS[key] = lib
    return _LIBS[key]


def _vp(x):
    return None if x is None else C.cast(x, C.c_void_p)


def _arr(a, dtype):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=dtype))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_error_model(lib_path: Optional[Path] = None) -> abi.ErrorModel:
    """oct_phmm_error_model_default: the reference's default_model_config (PCR-free, HiSeq-2500)."""
    m = abi.ErrorModel()
    load(lib_path).oct_phmm_error_model_default(C.byref(m))
    return m


class PinnedPool:
    """Page-locked host memory from the library (oct_phmm_host_alloc) as numpy arrays: inputs and `out` of big batches that live here go to and from the
    device without the library's staging copies. Freed by close()."""

    def __init__(self, lib_path: Optional[Path] = None):
        self.lib = load(lib_path)
        self.lib.oct_phmm_host_alloc.restype = C.c_void_p
        self.lib.oct_phmm_host_alloc.argtypes = [C.c_size_t]
        self.lib.oct_phmm_host_free.argtypes = [C.c_void_p]
        self.blocks = []

    def empty(self, n: int, dtype) -> np.ndarray:
        nbytes = max(int(n) * np.dtype(dtype).itemsize, 16)
        ptr = self.lib.oct_phmm_host_alloc(nbytes)
        if not ptr:
            raise MemoryError(f"oct_phmm_host_alloc({nbytes})")
        self.blocks.append(ptr)
        return np.frombuffer((C.c_char * nbytes).from_address(ptr), dtype=dtype, count=int(n))

    def copy(self, a: np.ndarray) -> np.ndarray:
        out = self.empty(a.size, a.dtype)
        out[:] = np.ascontiguousarray(a).reshape(-1)
        return out.reshape(a.shape)

    def batch(self, batch: "abi.Batch") -> "abi.Batch":
        """The same batch with every array in page-locked memory."""
        import copy as _copy
        b = _copy.copy(batch)
        for k, v in vars(batch).items():
            if isinstance(v, np.ndarray):
                setattr(b, k, self.copy(v))
        return b

    def close(self):
        for ptr in self.blocks:
            self.lib.oct_phmm_host_free(ptr)
        self.blocks = []


def test_set(name: str, value: Optional[str], lib_path: Optional[Path] = None) -> None:
    """oct_phmm_test_set: a test / A-B switch by its OCT_PHMM_* name (None removes it); read when a handle is created or a batch is uploaded."""
    code = load(lib_path).oct_phmm_test_set(name.encode(), None if value is None else str(value).encode())
    if code != abi.OK:
        raise EngineError(code, None, f"test_set({name})")


def error_model_by_name(library: Optional[str], sequencer: Optional[str], lib_path: Optional[Path] = None) -> abi.ErrorModel:
    """oct_phmm_error_model_by_name: one of the reference's built-in parameter sets (error_model_factory.cpp:220-517)."""
    m = abi.ErrorModel()
    lib = load(lib_path)
    lib.oct_phmm_error_model_by_name.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(abi.ErrorModel)]
    code = lib.oct_phmm_error_model_by_name(None if library is None else library.encode(), None if sequencer is None else sequencer.encode(), C.byref(m))
    if code != abi.OK:
        raise EngineError(code, None, f"error_model_by_name({library}, {sequencer})")
    return m


def error_model_by_label(label: str, lib_path: Optional[Path] = None) -> abi.ErrorModel:
    m = abi.ErrorModel()
    lib = load(lib_path)
    lib.oct_phmm_error_model_by_label.argtypes = [C.c_char_p, C.POINTER(abi.ErrorModel)]
    code = lib.oct_phmm_error_model_by_label(label.encode(), C.byref(m))
    if code != abi.OK:
        raise EngineError(code, None, f"error_model_by_label({label})")
    return m


def penalty_vectors(model: abi.ErrorModel, hap_bases: np.ndarray, hap_offsets: np.ndarray, substitution_mask=None, lib_path: Optional[Path] = None):
    """oct_phmm_penalty_vectors (host entry, no device): (gap_open, gap_extend, mask_fwd, prior_fwd, mask_rev, prior_rev), concatenated like the haplotypes."""
    lib = load(lib_path)
    bases = np.ascontiguousarray(hap_bases, dtype=np.uint8); off = np.ascontiguousarray(hap_offsets, dtype=np.uint32)
    n = int(off[-1]) if len(off) else 0
    go, ge, pf, pr = (np.zeros(max(n, 1), np.int8) for _ in range(4)); mf, mr = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint8)
    sub = None if substitution_mask is None else np.ascontiguousarray(substitution_mask, dtype=np.uint8)
    st = abi.Status()
    code = lib.oct_phmm_penalty_vectors(C.byref(model), len(off) - 1, _ptr(bases), _ptr(off), _ptr(sub), _ptr(go), _ptr(ge), _ptr(mf), _ptr(pf), _ptr(mr), _ptr(pr), C.byref(st))
    if code != abi.OK:
        raise EngineError(code, st, "penalty_vectors")
    return go[:n], ge[:n], mf[:n], pf[:n], mr[:n], pr[:n]


class _MotifPenalties(C.Structure):
    """oct_phmm_motif_penalties"""
    _fields_ = [("motif", C.c_char_p), ("motif_len", C.c_uint32), ("penalties", C.POINTER(C.c_int8)), ("n_penalties", C.c_uint32)]


class CustomIndelModel:
    """oct_phmm_custom_indel_model: the reference's CustomRepeatBasedIndelErrorModel - from a model file's text (oct_phmm_custom_indel_model_parse; EngineError EINVAL where the
    reference refuses the file) or from rows the caller holds (`open_rows` / `extend_rows`: {motif: penalties}; extend_rows None = no extension map)."""

    def __init__(self, text=None, *, open_rows=None, extend_rows=None, default_open=0, default_extend=3, lib_path: Optional[Path] = None):
        self.lib = load(lib_path)
        self.ptr = C.c_void_p()
        if text is not None:
            raw = text.encode() if isinstance(text, str) else bytes(text)
            code = self.lib.oct_phmm_custom_indel_model_parse(raw, len(raw), C.byref(self.ptr))
        else:
            keep = []

            def rows(d):
                arr = (_MotifPenalties * max(len(d), 1))()
                for i, (motif, pen) in enumerate(d.items()):
                    mb = motif.encode() if isinstance(motif, str) else bytes(motif)
                    pa = np.ascontiguousarray(pen, dtype=np.int8); keep.extend([mb, pa])
                    arr[i] = _MotifPenalties(mb, len(mb), pa.ctypes.data_as(C.POINTER(C.c_int8)), len(pa))
                return arr, len(d)
            oa, no = rows(open_rows or {})
            ea, ne = rows(extend_rows or {})
            code = self.lib.oct_phmm_custom_indel_model_create(oa, no, default_open, ea, ne, 0 if extend_rows is None else 1, default_extend, C.byref(self.ptr))
        if code != abi.OK:
            raise EngineError(code, None, "custom_indel_model")

    def info(self):
        do, de, no, ne, he = C.c_int8(), C.c_int8(), C.c_uint32(), C.c_uint32(), C.c_int32()
        self.lib.oct_phmm_custom_indel_model_info(self.ptr, C.byref(do), C.byref(de), C.byref(no), C.byref(ne), C.byref(he))
        return dict(default_open=do.value, default_extend=de.value, open_rows=no.value, extend_rows=ne.value, has_extend=bool(he.value))

    def close(self):
        if self.ptr:
            self.lib.oct_phmm_custom_indel_model_destroy(self.ptr); self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def custom_penalty_vectors(indel: CustomIndelModel, snv: Optional[abi.ErrorModel], hap_bases: np.ndarray, hap_offsets: np.ndarray, substitution_mask=None):
    """oct_phmm_custom_penalty_vectors (host entry, no device): gap vectors from the model file's rows, SNV vectors from `snv` (None: the default model, as the reference pairs them)."""
    lib = indel.lib
    bases = np.ascontiguousarray(hap_bases, dtype=np.uint8); off = np.ascontiguousarray(hap_offsets, dtype=np.uint32)
    n = int(off[-1]) if len(off) else 0
    go, ge, pf, pr = (np.zeros(max(n, 1), np.int8) for _ in range(4)); mf, mr = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint8)
    sub = None if substitution_mask is None else np.ascontiguousarray(substitution_mask, dtype=np.uint8)
    st = abi.Status()
    code = lib.oct_phmm_custom_penalty_vectors(indel.ptr, None if snv is None else C.byref(snv), len(off) - 1, _ptr(bases), _ptr(off), _ptr(sub),
                                               _ptr(go), _ptr(ge), _ptr(mf), _ptr(pf), _ptr(mr), _ptr(pr), C.byref(st))
    if code != abi.OK:
        raise EngineError(code, st, "custom_penalty_vectors")
    return go[:n], ge[:n], mf[:n], pf[:n], mr[:n], pr[:n]


class ResidentBatch:
    """A batch uploaded to HBM (oct_phmm_batch)."""

    def __init__(self, engine: "Engine", batch: abi.Batch):
        self.engine, self.batch = engine, batch
        self.ptr = C.c_void_p()
        st = abi.Status()
        r, h, g, f, p = batch.c_args()
        code = engine.lib.oct_phmm_batch_upload(engine.handle, _vp(r), _vp(h), _vp(g), _vp(f), _vp(p), C.byref(self.ptr), C.byref(st))
        if code != abi.OK:
            raise EngineError(code, st, "upload")

    def run(self):
        st = abi.Status()
        code = self.engine.lib.oct_phmm_batch_run(self.engine.handle, self.ptr, C.byref(st))
        if code != abi.OK:
            raise EngineError(code, st, "run")

    def wait(self):
        st = abi.Status()
        code = self.engine.lib.oct_phmm_batch_wait(self.engine.handle, self.ptr, C.byref(st))
        if code != abi.OK:
            raise EngineError(code, st, "wait")

    def download(self) -> np.ndarray:
        n = self.engine.lib.oct_phmm_batch_out_size(self.ptr)
        out = np.empty(max(n, 1), dtype=np.float64)
        st = abi.Status()
        code = self.engine.lib.oct_phmm_batch_download(self.engine.handle, self.ptr, _ptr(out), C.byref(st))
        if code != abi.OK:
            raise EngineError(code, st, "download")
        return out[:n]

    def genotype_likelihoods(self, sets) -> np.ndarray:
        """ConstantMixtureGenotypeLikelihoodModel::evaluate for vectors of genotypes on the resident matrix
        (oct_phmm_batch_genotype_likelihoods). sets: see abi.GenotypeSets.make."""
        gs, keep, n = abi.GenotypeSets.make(sets)
        out = np.empty(max(n, 1), dtype=np.float64)
        st = abi.Status()
        code = self.engine.lib.oct_phmm_batch_genotype_likelihoods(self.engine.handle, self.ptr, C.byref(gs), _ptr(out), C.byref(st))
        if code != abi.OK:
            raise EngineError(code, st, "genotype_likelihoods")
        return out[:n]

    def candidate_positions(self) -> list:
        """Test seam (oct_phmm_batch_candidate_positions): per (haplotype, read) pair, in the order of Batch.read_pairs(), the list of candidate
        mapping positions the last run used."""
        n, S = self.batch.n_read_pairs(), int(self.engine.cfg.max_mapping_positions)
        cnt, pos = np.zeros(max(n, 1), np.uint8), np.zeros(max(n * S, 1), np.uint32)
        st = abi.Status()
        code = self.engine.lib.oct_phmm_batch_candidate_positions(self.engine.handle, self.ptr, _ptr(cnt), _ptr(pos), C.byref(st))
        if code != abi.OK:
            raise EngineError(code, st, "candidate_positions")
        return [pos[e * S: e * S + int(cnt[e])].tolist() for e in range(n)]

    def penalty_vectors(self):
        """Test seam (oct_phmm_batch_penalty_vectors): the six per-haplotype vectors this batch runs with (given by the caller or generated at upload)."""
        n = int(self.batch.hap_offsets[-1])
        go, ge, pf, pr = (np.zeros(max(n, 1), np.int8) for _ in range(4)); mf, mr = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint8)
        st = abi.Status()
        code = self.engine.lib.oct_phmm_batch_penalty_vectors(self.engine.handle, self.ptr, _ptr(go), _ptr(ge), _ptr(mf), _ptr(pf), _ptr(mr), _ptr(pr), C.byref(st))
        if code != abi.OK:
            raise EngineError(code, st, "batch_penalty_vectors")
        return go[:n], ge[:n], mf[:n], pf[:n], mr[:n], pr[:n]

    def device_sized(self) -> bool:
        """oct_phmm_batch_device_sized: the step runs without the mid-step read-back of the task counts."""
        return bool(self.engine.lib.oct_phmm_batch_device_sized(self.ptr))

    def stats(self) -> dict:
        s = abi.Stats()
        self.engine.lib.oct_phmm_batch_stats(self.ptr, C.byref(s))
        return s.as_dict()

    def kernel_time(self):
        ms, n = C.c_double(0), C.c_uint32(0)
        self.engine.lib.oct_phmm_batch_kernel_time(self.ptr, C.byref(ms), C.byref(n))
        return ms.value, n.value

    def kernel_time_by_kind(self) -> dict:
        """{kind: (ms, launches)} of the last run's DP launches; kinds as in oct_phmm.h."""
        ms, n = (C.c_double * 4)(), (C.c_uint32 * 4)()
        self.engine.lib.oct_phmm_batch_kernel_time_by_kind(self.ptr, C.byref(ms), C.byref(n))
        return {k: (ms[i], n[i]) for i, k in enumerate(("score_fast", "trace_fast", "score_generic", "trace_generic"))}

    def free(self):
        if self.ptr:
            self.engine.lib.oct_phmm_batch_free(self.engine.handle, self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """One oct_phmm_handle (= one HaplotypeLikelihoodModel configuration on one device)."""

    def __init__(self, cfg: Optional[abi.Config] = None, lib_path: Optional[Path] = None):
        self.lib = load(lib_path)
        self.cfg = cfg if cfg is not None else abi.Config.default()
        self.handle = C.c_void_p()
        code = self.lib.oct_phmm_create(C.byref(self.cfg), C.byref(self.handle))
        if code != abi.OK:
            raise EngineError(code, None, "create: " + self.lib.oct_phmm_strerror(code).decode())

    @property
    def band_size(self) -> int:
        return self.lib.oct_phmm_band_size(self.handle)

    def close(self):
        if self.handle:
            self.lib.oct_phmm_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def populate(self, batch: abi.Batch, raise_on_error: bool = True, out: Optional[np.ndarray] = None):
        """HaplotypeLikelihoodArray::populate. Returns (out, status). `out`: optional preallocated float64 buffer."""
        if out is None:
            out = np.full(max(batch.out_size(), 1), np.nan, dtype=np.float64)
        assert out.dtype == np.float64 and out.size >= batch.out_size() and out.flags.c_contiguous
        st = abi.Status()
        r, h, g, f, p = batch.c_args()
        code = self.lib.oct_phmm_populate(self.handle, _vp(r), _vp(h), _vp(g), _vp(f), _vp(p), _ptr(out), C.byref(st))
        if code != abi.OK and raise_on_error:
            raise EngineError(code, st, "populate")
        return out[:batch.out_size()], st

    def align(self, batch: abi.Batch, max_cigar_ops: int = 64, raise_on_error: bool = True):
        """HaplotypeLikelihoodModel::align for every (haplotype, read) pair (oct_phmm_align). Returns (result dict, status)."""
        n = batch.n_read_pairs()
        out, arrays = abi.Alignments.make(n, max_cigar_ops)
        st = abi.Status()
        r, h, g, f, p = batch.c_args()
        code = self.lib.oct_phmm_align(self.handle, _vp(r), _vp(h), _vp(g), _vp(f), _vp(p), C.byref(out), C.byref(st))
        if code != abi.OK and raise_on_error:
            raise EngineError(code, st, "align")
        return abi.alignments_result(arrays, n, max_cigar_ops), st

    def align_candidate_counts(self, n_pairs: int):
        """oct_phmm_align_candidate_counts: (candidates the device mapper kept per pair in the last align call, pairs whose list filled every slot)."""
        counts = np.zeros(max(n_pairs, 1), np.uint8)
        n_sat = C.c_uint32(0)
        code = self.lib.oct_phmm_align_candidate_counts(self.handle, _ptr(counts), n_pairs, C.byref(n_sat))
        if code != abi.OK:
            raise EngineError(code, None, "align_candidate_counts")
        return counts[:n_pairs], int(n_sat.value)

    def set_error_model(self, model: Optional[abi.ErrorModel]):
        """oct_phmm_set_error_model: batches whose six penalty vectors are None (Batch.without_penalty_vectors) get them generated at upload."""
        code = self.lib.oct_phmm_set_error_model(self.handle, None if model is None else C.byref(model))
        if code != abi.OK:
            raise EngineError(code, None, "set_error_model")

    def set_custom_error_model(self, indel: "CustomIndelModel", snv: Optional[abi.ErrorModel] = None):
        """oct_phmm_set_custom_error_model: NULL-vector batches get their gap vectors from a model file's rows (set_error_model takes it away again)."""
        code = self.lib.oct_phmm_set_custom_error_model(self.handle, indel.ptr, None if snv is None else C.byref(snv))
        if code != abi.OK:
            raise EngineError(code, None, "set_custom_error_model")

    def probe_clock(self, window_ms: float = 5.0) -> float:
        """oct_phmm_probe_clock: shader clock (GHz) sustained over the window, beside whatever else runs on the device."""
        g = C.c_double(0)
        code = self.lib.oct_phmm_probe_clock(self.handle, float(window_ms), C.byref(g))
        if code != abi.OK:
            raise EngineError(code, None, "probe_clock")
        return g.value

    def set_timing(self, enabled: bool = True):
        """Bracket the DP launches with HIP events (ResidentBatch.kernel_time*); off by default."""
        self.lib.oct_phmm_set_timing(self.handle, 1 if enabled else 0)

    def upload(self, batch: abi.Batch) -> ResidentBatch:
        return ResidentBatch(self, batch)

    def align_windows(self, truths, targets, quals, gap_open, gap_extend=None, gap_extend_scalar=1, snv_mask=None,
                      snv_prior=None, nuc_prior=2, traceback=False, lhs_flank=None, rhs_flank=None):
        """simd::PairHMM::align on explicit windows (test seam). Lists of bytes / arrays, one per window."""
        n = len(truths)
        cat = lambda xs, dt: (np.concatenate([np.frombuffer(bytes(x), np.uint8) if isinstance(x, (bytes, bytearray))
                                              else np.asarray(x) for x in xs]).astype(dt) if n else np.zeros(0, dt))
        off = lambda xs: np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.uint32)
        tr, toff = cat(truths, np.uint8), off(truths)
        tg, goff = cat(targets, np.uint8), off(targets)
        q = cat(quals, np.uint8)
        go = cat(gap_open, np.int8)
        ge = None if gap_extend is None else cat(gap_extend, np.int8)
        m = None if snv_mask is None else cat(snv_mask, np.uint8)
        pr = None if snv_prior is None else cat(snv_prior, np.int8)
        B = self.band_size
        scores = np.zeros(n, np.int32)
        fp = np.zeros(n, np.int32)
        aoff = np.concatenate([[0], np.cumsum([2 * (len(t) + B) + 1 for t in targets])]).astype(np.uint32)
        a1 = np.zeros(int(aoff[-1]) + 1, np.uint8)
        a2 = np.zeros(int(aoff[-1]) + 1, np.uint8)
        lf = None if lhs_flank is None else _arr(lhs_flank, np.int32)
        rf = None if rhs_flank is None else _arr(rhs_flank, np.int32)
        fl, ms = np.zeros(n, np.int32), np.zeros(n, np.int32)
        st = abi.Status()
        self.lib.oct_phmm_align_windows.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 5 + [C.c_void_p, C.c_void_p, C.c_int32,
                                                    C.c_void_p, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 10
        code = self.lib.oct_phmm_align_windows(self.handle, n, _ptr(tr), _ptr(toff), _ptr(tg), _ptr(q), _ptr(goff), _ptr(go), _ptr(ge),
                                               int(gap_extend_scalar), _ptr(m), _ptr(pr), int(nuc_prior), 1 if traceback else 0,
                                               _ptr(scores), _ptr(fp), _ptr(a1), _ptr(a2), _ptr(aoff), _ptr(lf), _ptr(rf),
                                               _ptr(fl) if lf is not None else None, _ptr(ms) if lf is not None else None, C.byref(st))
        if code != abi.OK:
            raise EngineError(code, st, "align_windows")
        res = []
        for i in range(n):
            d = dict(score=int(scores[i]))
            if traceback:
                s1 = bytes(a1[aoff[i]:aoff[i + 1]]).split(b"\0")[0].decode("latin1")
                s2 = bytes(a2[aoff[i]:aoff[i + 1]]).split(b"\0")[0].decode("latin1")
                d.update(first_pos=int(fp[i]), align1=s1 if fp[i] >= 0 else "", align2=s2 if fp[i] >= 0 else "")
                if lf is not None:
                    d.update(flank_score=int(fl[i]), mask_size=int(ms[i]))
            res.append(d)
        return res


class Server:
    """oct_phmm_server: one device queue shared by many calling threads; concurrent single-region populate calls are answered by
    multi-region batches. `populate` is thread-safe and blocks until its own result is ready."""

    def __init__(self, cfg: Optional[abi.Config] = None, max_regions_per_batch: int = 0, lib_path: Optional[Path] = None,
                 devices: Optional[list] = None):
        """devices: GPU ordinals to serve from (oct_phmm_server_create_multi); None = cfg.device_id only."""
        self.lib = load(lib_path)
        self.cfg = cfg if cfg is not None else abi.Config.default()
        self.ptr = C.c_void_p()
        self.n_devices = 1 if devices is None else len(devices)
        if devices is None:
            code = self.lib.oct_phmm_server_create(C.byref(self.cfg), int(max_regions_per_batch), C.byref(self.ptr))
        else:
            ids = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            code = self.lib.oct_phmm_server_create_multi(C.byref(self.cfg), ids, len(devices), int(max_regions_per_batch), C.byref(self.ptr))
        if code != abi.OK:
            raise EngineError(code, None, "server_create")

    def populate(self, batch: abi.Batch, raise_on_error: bool = True, out: Optional[np.ndarray] = None):
        if out is None:
            out = np.full(max(batch.out_size(), 1), np.nan, dtype=np.float64)
        st = abi.Status()
        r, h, g, f, p = batch.c_args()
        assert g is None, "one region per call"
        code = self.lib.oct_phmm_server_populate(self.ptr, _vp(r), _vp(h), _vp(f), _vp(p), _ptr(out), C.byref(st))
        if code != abi.OK and raise_on_error:
            raise EngineError(code, st, "server_populate")
        return out[:batch.out_size()], st

    def set_error_model(self, model: Optional[abi.ErrorModel]):
        code = self.lib.oct_phmm_server_set_error_model(self.ptr, None if model is None else C.byref(model))
        if code != abi.OK:
            raise EngineError(code, None, "server_set_error_model")

    def set_custom_error_model(self, indel: "CustomIndelModel", snv: Optional[abi.ErrorModel] = None):
        code = self.lib.oct_phmm_server_set_custom_error_model(self.ptr, indel.ptr, None if snv is None else C.byref(snv))
        if code != abi.OK:
            raise EngineError(code, None, "server_set_custom_error_model")

    def stats(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self.lib.oct_phmm_server_stats(self.ptr, C.byref(a), C.byref(b))
        return a.value, b.value

    def device_calls(self) -> list:
        a = (C.c_uint64 * self.n_devices)()
        self.lib.oct_phmm_server_device_calls(self.ptr, a, self.n_devices)
        return [int(x) for x in a]

    def close(self):
        if self.ptr:
            self.lib.oct_phmm_server_destroy(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
