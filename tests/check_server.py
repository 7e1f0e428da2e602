"""oct_phmm_server: concurrent single-region calls from several threads are answered by multi-region batches; every caller must get exactly
what its own oct_phmm_populate call would have returned (values, or its own error status)."""
import threading

import numpy as np

import oracle
from backends import build_sim, make_engine
from check_populate import mapper_positions
from octopus_amd import abi, engine, synth


def make_requests(rng, n, band=8):
    reqs = []
    for i in range(n):
        T, Lh = int(rng.integers(30, 70)), int(rng.integers(150, 260))
        g = synth.make_region(rng, int(rng.integers(4, 30)), int(rng.integers(1, 6)), T=T, Lh=Lh, B=band,
                              flank=None if i % 4 == 0 else (int(rng.integers(0, 40)), int(rng.integers(0, 40))), positions="none")
        if i % 7 == 3:                                                 # a region whose haplotypes cannot hold its reads: ShortHaplotypeError for this caller only
            g["haps"] = [h[:T + 2 * band - 4] for h in g["haps"]]
            g["begin"] = np.minimum(g["begin"], 3)
        b = synth.batch_from_regions([g])
        if i % 5 == 1:                                                 # templates
            R = g["reads"].shape[0]; rows, r = [0], 0
            while r < R:
                r += 2 if (R - r >= 2 and rng.random() < 0.5) else 1
                rows.append(r)
            b.row_offsets = np.asarray(rows, np.uint32)
        if i % 6 == 2:                                                 # caller-provided candidate positions: served on its own
            mapper_positions(b, rng=rng, junk=0.3)
        reqs.append(b)
    return reqs


def check_server(backend, n_threads=5, per_thread=7, seed=17, band=8, devices=None):
    """devices: serve through oct_phmm_server_create_multi on these GPU ordinals (the same ordinal may repeat: each entry gets its own
    workers and handles, which is how a one-GPU box exercises the several-device code path)."""
    rng = np.random.default_rng(seed)
    reqs = [make_requests(rng, per_thread, band) for _ in range(n_threads)]
    cfg = abi.Config.default(max_indel_error=band)
    lib_path = build_sim() if backend == "sim" else None
    srv = engine.Server(cfg, lib_path=lib_path, devices=devices)
    got = [[None] * per_thread for _ in range(n_threads)]
    errors = []

    def worker(t):
        try:
            for rep in range(2):
                for i, b in enumerate(reqs[t]):
                    out, st = srv.populate(b, raise_on_error=False)
                    got[t][i] = (out.copy(), st.code, st.hap_index, st.read_index, st.required_extension)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    [t.start() for t in ths]; [t.join() for t in ths]
    calls, batches = srv.stats()
    by_device = srv.device_calls()
    srv.close()
    assert sum(by_device) == calls and len(by_device) == (1 if devices is None else len(devices))
    assert not errors, errors
    assert calls == 2 * n_threads * per_thread and 0 < batches <= calls
    n_err = 0
    for t in range(n_threads):
        for i, b in enumerate(reqs[t]):
            want, wst, _ = oracle.populate(cfg, b)
            out, code, hi, ri, ext = got[t][i]
            assert code == wst.code, (t, i, code, wst.code)
            if code == abi.OK:
                assert np.max(np.abs(out - want), initial=0.0) <= (0.0 if backend == "sim" else 1e-9)
            else:
                n_err += 1
                assert (hi, ri, ext) == (wst.hap_index, wst.read_index, wst.required_extension)
    assert n_err > 0
    return calls, batches


def check_server_rejects_malformed_calls(backend):
    """A call with a NULL array (or no output buffer) gets OCT_PHMM_EINVAL back from oct_phmm_server_populate itself and never reaches a
    worker thread; the server keeps answering the well-formed calls around it."""
    import ctypes as C
    rng = np.random.default_rng(5)
    good = make_requests(rng, 2, 8)[0]
    cfg = abi.Config.default(max_indel_error=8)
    srv = engine.Server(cfg, lib_path=build_sim() if backend == "sim" else None)
    want, _ = srv.populate(good)
    out = np.zeros(good.out_size())
    fl = good.c_flank()
    for field, owner in (("qualities", 0), ("offsets", 0), ("gap_open", 1), ("snv_mask_rev", 1), (None, None)):
        r, h = good.c_reads(), good.c_haps()
        if field is not None:
            setattr((r, h)[owner], field, None)
        st = abi.Status()
        code = srv.lib.oct_phmm_server_populate(srv.ptr, C.cast(C.byref(r), C.c_void_p), C.cast(C.byref(h), C.c_void_p),
                                                None if fl is None else C.cast(C.byref(fl), C.c_void_p), None,
                                                out.ctypes.data_as(C.c_void_p) if field is not None else None, C.byref(st))
        assert code == abi.EINVAL and st.code == abi.EINVAL, (field, code)
    again, _ = srv.populate(good)
    assert np.array_equal(again, want)
    srv.close()


def check_server_contract_violation_reaches_only_its_caller(backend):
    """The callers' threads collect what an upload must know about their region's bytes (round 5): a call whose read carries a quality above 127, or whose haplotype carries a negative
    penalty, makes the device batch it rides in fail its upload - the batch is then answered call by call, the offender gets OCT_PHMM_EINVAL from its own populate and every other caller its matrix."""
    rng = np.random.default_rng(23)
    reqs = [r for r in make_requests(rng, 14, 8) if r.pos_offsets is None]
    import copy
    bad_q = copy.copy(reqs[2]); bad_q._keep = []; bad_q.read_quals = reqs[2].read_quals.copy(); bad_q.read_quals[3] = 200
    bad_p = copy.copy(reqs[5]); bad_p._keep = []; bad_p.gap_open = reqs[5].gap_open.copy(); bad_p.gap_open[1] = -3
    calls = list(reqs); calls[2] = bad_q; calls[5] = bad_p
    cfg = abi.Config.default(max_indel_error=8)
    srv = engine.Server(cfg, lib_path=build_sim() if backend == "sim" else None)
    got = [None] * len(calls)

    def worker(t):
        for i in range(t, len(calls), 4):
            out, st = srv.populate(calls[i], raise_on_error=False)
            got[i] = (out.copy(), st.code)
    for rep in range(2):
        ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
        [t.start() for t in ths]; [t.join() for t in ths]
        for i, b in enumerate(calls):
            out, code = got[i]
            if i in (2, 5):
                assert code == abi.EINVAL, (i, code)
                continue
            want, wst, _ = oracle.populate(cfg, b)
            assert code == wst.code, (i, code, wst.code)
            if code == abi.OK:
                assert np.max(np.abs(out - want), initial=0.0) <= (0.0 if backend == "sim" else 1e-9)
    srv.close()
    return True
