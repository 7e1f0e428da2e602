"""The product's per-haplotype penalty vectors (octopus_amd/csrc/phmm_error_model.hpp behind oct_phmm_penalty_vectors, and
generated inside populate when the caller leaves the six vectors NULL) against the REFERENCE's own error-model classes
(BasicRepeatBasedIndelErrorModel / BasicRepeatBasedSNVErrorModel over its tandem library, compiled in place: oracle/_ref) and the oracle."""
import os

import numpy as np

import oracle
from backends import build_sim, make_engine
from octopus_amd import abi, engine, synth
from test_oracle_error_models import model as default_tables, random_sequence, ref_penalty_vectors

NAMES = ("gap_open", "gap_extend", "mask_fwd", "prior_fwd", "mask_rev", "prior_rev")


def corpus(seed, n_strings, with_sub=True):
    """Random haplotypes with planted repeats of periods 1-6 (the corpus of tests/test_oracle_error_models.py), some with non-ACGT bases,
    some with substitution masks; concatenated like a batch's haplotypes."""
    rng = np.random.default_rng(seed)
    seqs, subs = [], []
    for it in range(n_strings):
        seq = random_sequence(rng, int(rng.integers(2, 420)), b"ACGT" if it % 6 else b"ACGTN")
        sub = np.zeros(len(seq), np.uint8)
        if with_sub and it % 3 == 0:
            for _ in range(int(rng.integers(0, 4))):
                a = int(rng.integers(0, len(seq))); sub[a:a + int(rng.integers(1, 6))] = 1
        seqs.append(seq); subs.append(sub)
    bases = np.frombuffer(b"".join(seqs), np.uint8)
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint32)
    return seqs, subs, bases, off


def check_host_entry(lib_path=None, n_strings=2400, reference=True):
    """oct_phmm_penalty_vectors (device-free) on the 2,400-string corpus: all six vectors equal the reference's classes', string by string."""
    m = engine.default_error_model(lib_path)
    assert bytes(m) == bytes(default_tables())                  # oct_phmm_error_model_default = the factory's default tables, expanded
    seqs, subs, bases, off = corpus(77, n_strings)
    got = engine.penalty_vectors(m, bases, off, np.concatenate(subs), lib_path=lib_path)
    n_diff = 0
    for i, (seq, sub) in enumerate(zip(seqs, subs)):
        want = ref_penalty_vectors(seq, sub) if reference else oracle.penalty_vectors(m, seq, sub)
        for name, g, w in zip(NAMES, got, want):
            assert np.array_equal(g[off[i]:off[i + 1]], w), (name, i, seq, np.flatnonzero(g[off[i]:off[i + 1]] != w)[:5])
        n_diff += 1
    # degenerate inputs: no haplotypes, one base, no SNV model
    assert all(len(v) == 0 for v in engine.penalty_vectors(m, np.zeros(0, np.uint8), np.zeros(1, np.uint32), lib_path=lib_path))
    m2 = engine.default_error_model(lib_path); m2.use_snv_model = 0
    s = np.frombuffer(b"ACGTTTTTTTTGA", np.uint8)
    go, ge, mf, pf, mr, pr = engine.penalty_vectors(m2, s, np.asarray([0, len(s)], np.uint32), lib_path=lib_path)
    assert np.array_equal(mf, s) and np.array_equal(mr, s) and set(pf.tolist()) == {100} and set(pr.tolist()) == {100}      # model.cpp:68-73
    return n_diff


def check_populate_generates_the_vectors(backend, tol=0.0, where=("host", "device")):
    """A batch uploaded WITHOUT its six vectors (all pointers NULL) after oct_phmm_set_error_model: the vectors the library generated equal
    the reference's (oct_phmm_batch_penalty_vectors), and the matrix equals the one computed from given vectors - on both generation paths."""
    rng = np.random.default_rng(31)
    m = engine.default_error_model(build_sim() if backend == "sim" else None)
    old = os.environ.get("OCT_PHMM_PENALTIES")
    n = 0
    try:
        for path in where:
            os.environ["OCT_PHMM_PENALTIES"] = path
            for B, R, H, T, Lh in ((8, 14, 5, 50, 160), (16, 10, 4, 60, 200)):
                g = synth.make_region(rng, R, H, T=T, Lh=Lh, B=B, flank=(20, 20), positions="none")
                for h in g["haps"][1:3]:                      # plant repeats so that the vectors are not flat
                    a = int(rng.integers(30, Lh - 60)); h[a:a + 14] = ord("A"); h[a + 20:a + 36] = np.frombuffer(b"CG" * 8, np.uint8)
                batch = synth.batch_from_regions([g])
                want_vec = [np.concatenate(v) for v in zip(*[ref_penalty_vectors(bytes(h)) for h in g["haps"]])]
                given = synth.batch_from_regions([g])
                given.gap_open, given.gap_extend, given.snv_mask_fwd, given.snv_prior_fwd, given.snv_mask_rev, given.snv_prior_rev = want_vec
                eng = make_engine(backend, max_indel_error=B)
                want, _ = eng.populate(given)
                eng.set_error_model(m)
                rb = eng.upload(batch.without_penalty_vectors())
                for name, gvec, wvec in zip(NAMES, rb.penalty_vectors(), want_vec):
                    assert np.array_equal(gvec, wvec), (path, name)
                rb.run(); got = rb.download().copy(); rb.free()
                assert np.max(np.abs(got - want), initial=0.0) <= tol
                one_shot, _ = eng.populate(batch.without_penalty_vectors())
                assert np.array_equal(one_shot, got)
                eng.set_error_model(None)                      # without a model NULL vectors are an error, not a crash
                _, st = eng.populate(batch.without_penalty_vectors(), raise_on_error=False)
                assert st.code == abi.EINVAL
                eng.close()
                n += got.size
    finally:
        if old is None:
            os.environ.pop("OCT_PHMM_PENALTIES", None)
        else:
            os.environ["OCT_PHMM_PENALTIES"] = old
    return n


def check_device_kernels_on_the_corpus(backend, n_strings=600, where=("device", "lanes"), seed=78):
    """The two device kernels (one wave per haplotype with its workspace in LDS; one lane per haplotype with its workspace in HBM) on corpus
    haplotypes with substitution masks: the vectors a NULL-vector upload leaves on the device equal the device-free host entry's, which
    check_host_entry pins to the reference's classes."""
    lib_path = build_sim() if backend == "sim" else None
    m = engine.default_error_model(lib_path)
    seqs, subs, bases, off = corpus(seed, n_strings)
    sub = np.concatenate(subs)
    want = engine.penalty_vectors(m, bases, off, sub, lib_path=lib_path)
    rng = np.random.default_rng(seed)
    read = rng.choice(np.frombuffer(b"ACGT", np.uint8), 30)
    batch = abi.Batch(read_bases=read, read_quals=np.full(30, 30, np.uint8), read_offsets=np.asarray([0, 30], np.uint32), mapq=np.asarray([60], np.uint8),
                      reverse=np.zeros(1, np.uint8), read_ref_begin=np.zeros(1, np.int64), row_offsets=None, hap_bases=bases.copy(), hap_offsets=off,
                      hap_ref_begin=np.zeros(len(seqs), np.int64), gap_open=None, gap_extend=None, snv_mask_fwd=None, snv_prior_fwd=None,
                      snv_mask_rev=None, snv_prior_rev=None)
    old = os.environ.get("OCT_PHMM_PENALTIES")
    try:
        for path in where:
            os.environ["OCT_PHMM_PENALTIES"] = path
            eng = make_engine(backend, max_indel_error=8)
            eng.set_error_model(m)
            batch.substitution_mask = sub                           # oct_phmm_haplotypes::substitution_mask: travels with the call
            rb = eng.upload(batch)
            got = rb.penalty_vectors()
            rb.free(); eng.close()
            for name, g, w in zip(NAMES, got, want):
                bad = np.flatnonzero(g != w)
                assert bad.size == 0, (path, name, int(np.searchsorted(off, bad[0], "right") - 1), bad[:5])
    finally:
        if old is None:
            os.environ.pop("OCT_PHMM_PENALTIES", None)
        else:
            os.environ["OCT_PHMM_PENALTIES"] = old
    return len(seqs)


def check_align_and_server_generate_the_vectors(backend, tol=0.0):
    """oct_phmm_align and the region server with NULL penalty vectors + a model: same alignments / matrices as with the vectors given."""
    import threading
    from check_server import make_requests
    lib_path = build_sim() if backend == "sim" else None
    m = engine.default_error_model(lib_path)
    rng = np.random.default_rng(91)

    def with_model_vectors(batch, model=None):
        vec = engine.penalty_vectors(model if model is not None else m, batch.hap_bases, batch.hap_offsets, getattr(batch, "substitution_mask", None), lib_path=lib_path)
        import copy
        g = copy.copy(batch); g._keep = []
        g.gap_open, g.gap_extend, g.snv_mask_fwd, g.snv_prior_fwd, g.snv_mask_rev, g.snv_prior_rev = vec
        return g

    # align
    b = synth.batch_from_regions([synth.make_region(rng, 25, 4, T=60, Lh=180, B=8, flank=(20, 20), positions="none", indels_per_read=1)])
    eng = make_engine(backend, max_indel_error=8)
    want, wst = eng.align(with_model_vectors(b), 64)
    eng.set_error_model(m)
    got, gst = eng.align(b.without_penalty_vectors(), 64)
    eng.close()
    assert wst.code == gst.code == abi.OK
    assert got["cigar_strings"] == want["cigar_strings"] and np.array_equal(got["mapping_position"], want["mapping_position"])
    assert np.max(np.abs(got["likelihood"] - want["likelihood"]), initial=0.0) <= tol
    # region server: callers with and without vectors at the same time
    # ... and every third caller's haplotypes carry substitutions (oct_phmm_haplotypes::substitution_mask travels with the call into the server's batches)
    reqs = [r for r in make_requests(rng, 12, 8) if r.pos_offsets is None]
    for i, r in enumerate(reqs):
        if i % 3 == 0:
            r.substitution_mask = (rng.random(int(r.hap_offsets[-1])) < 0.15).astype(np.uint8)
    cfg = abi.Config.default(max_indel_error=8)
    m2 = engine.default_error_model(lib_path)                          # a second model, installed while the server is up
    for k in range(len(m2.at_homopolymer_open)):
        m2.at_homopolymer_open[k] = max(3, m2.at_homopolymer_open[k] - 2); m2.dinucleotide_extend[k] = 4
    srv = engine.Server(cfg, lib_path=lib_path)
    for model in (m, m2):
        ref_eng = make_engine(backend, max_indel_error=8)
        want = [ref_eng.populate(with_model_vectors(r, model), raise_on_error=False) for r in reqs]
        want = [(o.copy(), st.code) for o, st in want]
        ref_eng.close()
        srv.set_error_model(model)
        got = [None] * len(reqs)
        errors = []

        def worker(t):
            try:
                for i in range(t, len(reqs), 3):
                    r = reqs[i].without_penalty_vectors() if i % 2 == 0 else with_model_vectors(reqs[i], model)
                    o, st = srv.populate(r, raise_on_error=False)
                    got[i] = (o.copy(), st.code)
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        ths = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
        [t.start() for t in ths]; [t.join() for t in ths]
        assert not errors, errors
        for (go, gc), (wo, wc) in zip(got, want):
            assert gc == wc
            if gc == abi.OK:
                assert np.max(np.abs(go - wo), initial=0.0) <= tol
    srv.close()
    return len(reqs)


# ---- the model read from a file: CustomRepeatBasedIndelErrorModel (oct_phmm_custom_indel_model_*) ------------------------------------------------------------------
def ref_custom_indel(text: bytes, seq: bytes):
    """The reference's own make_penalty_map + CustomRepeatBasedIndelErrorModel::set_penalties (oracle/ref_errmodel_bridge.cpp): (rc, gap_open, gap_extend)."""
    import ctypes as C
    n = len(seq)
    go, ge = np.zeros(max(n, 1), np.int8), np.zeros(max(n, 1), np.int8)
    f = oracle.ref().ref_custom_indel_penalties
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_uint32, C.c_void_p, C.c_void_p]; f.restype = C.c_int
    rc = f(text, len(text), seq, n, go.ctypes.data, ge.ctypes.data)
    return rc, go[:n], ge[:n]


MOTIFS = [b"A", b"C", b"G", b"T", b"N", b"AC", b"CA", b"CG", b"GC", b"AT", b"TA", b"NN", b"ACG", b"CAG", b"AAT", b"NNN", b"AAAC", b"ACGT", b"NNNN", b"ACGTA", b"AAAAC", b"NNNNN",
          b"a", b"AN", b"ACGTAC", b"NNNNNNNNNN"]


def random_model_text(rng, well_formed=True):
    """A model file: open rows, maybe '+' rows, comments, blank lines, with or without the final newline; ill-formed ones carry one defect."""
    lines = []
    n_open = int(rng.integers(1, 9))
    for m in rng.choice(len(MOTIFS), n_open, replace=bool(rng.random() < 0.2)):
        lines.append(MOTIFS[int(m)] + b":" + b",".join(b"%d" % int(v) for v in rng.integers(0, 60, int(rng.integers(1, 24)))))
    if rng.random() < 0.6:
        for m in rng.choice(len(MOTIFS), int(rng.integers(1, 6)), replace=False):
            lines.append(MOTIFS[int(m)] + b"+:" + b",".join(b"%d" % int(v) for v in rng.integers(0, 20, int(rng.integers(1, 12)))))
    rng.shuffle(lines)
    for _ in range(int(rng.integers(0, 3))):
        lines.insert(int(rng.integers(0, len(lines) + 1)), b"# a comment: with, a colon" if rng.random() < 0.7 else b"")
    text = b"\n".join(lines) + (b"\n" if rng.random() < 0.7 else b"")
    if not well_formed:
        defect = int(rng.integers(0, 8))
        body = text.rstrip(b"\n")
        text = [body + b"\nA:\n", body + b"\nAC:1,,2\n", body + b"\nA:1, 2\n", body.replace(b"\n", b"\r\n") + b"\r\n", body + b"\nC:x\n", body + b"\nG:128\n", body + b"\n:4,4\n",
                body + b"\nACG"][defect]
    return text


FIXED_TEXTS = [(b"", False), (b"# nothing but a comment\n", False), (b"\n\n", False), (b":1,2\n", False), (b"A:\n", False), (b"A:", False), (b"A:1,,2\n", False), (b"A:1, 2\n", False),
               (b"A:1\r\n", False), (b"A:x\n", False), (b"A:300\n", False), (b"A:-129\n", False), (b"A:-128,127\n", True), (b"+:1\n", False), (b"A+:1\n", False), (b"A", False),
               (b"A:1,2", True), (b"A:1,\n", False), (b"A:1,", True), (b"A:+5,-3\n", True), (b"A:1\n\n\nC:2\n", True), (b"A\nB:1\n", True), (b"A:99999999999\n", False),
               (b"A:2147483648\n", False), (b"A:+\n", False), (b"A:-\n", False), (b"A:1\nA:9\n", True), (b"N:7\nNN:6\nA+:2\nN+:1,1,4\n", True), (b"#c\nAC:5\n#d", True), (b"A:1\n#", True),
               (b"A:007\n", True), (b"A: \n", False), (b"A:1.5\n", False)]


def check_custom_model_file(lib_path=None, n_models=70, n_strings=24):
    """Model files read by the product and by the reference's own make_penalty_map: the same texts accepted and refused; for the accepted ones gap-open and gap-extension
    vectors equal CustomRepeatBasedIndelErrorModel's on corpus strings (motif rows, N rows, the iteration-order default), the four SNV vectors equal the default model's."""
    rng = np.random.default_rng(2025)
    seqs, subs, bases, off = corpus(5, n_strings)
    seqs = list(seqs) + [b"", b"A", b"ACGT", b"ACGTTGCAAGTC"]                       # ... and strings without any repeat: the two defaults everywhere
    bases = np.frombuffer(b"".join(seqs), np.uint8); off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint32)
    snv_want = engine.penalty_vectors(engine.default_error_model(lib_path), bases, off, lib_path=lib_path)[2:]
    texts = [(t, ok) for t, ok in FIXED_TEXTS] + [(random_model_text(rng, True), None) for _ in range(n_models)] + [(random_model_text(rng, False), None) for _ in range(n_models // 2)]
    n_ok = n_bad = n_default_hits = 0
    for text, expect_ok in texts:
        rc_ref = ref_custom_indel(text, b"ACGT")[0]
        try:
            model = engine.CustomIndelModel(text, lib_path=lib_path)
        except engine.EngineError as e:
            assert e.code == abi.EINVAL and rc_ref != 0, (text, rc_ref)
            assert expect_ok in (None, False), text
            n_bad += 1
            continue
        assert rc_ref == 0, (text, rc_ref)
        assert expect_ok in (None, True), text
        got = engine.custom_penalty_vectors(model, None, bases, off)
        info = model.info()
        for i, seq in enumerate(seqs):
            _, go, ge = ref_custom_indel(text, seq)
            assert np.array_equal(got[0][off[i]:off[i + 1]], go), ("gap_open", text, seq)
            assert np.array_equal(got[1][off[i]:off[i + 1]], ge), ("gap_extend", text, seq)
            n_default_hits += int(np.count_nonzero(go == info["default_open"]))
        for g, w in zip(got[2:], snv_want):
            assert np.array_equal(g, w)
        # the same model from rows + stated defaults (what a caller holding the reference's two maps passes)
        open_rows, extend_rows = {}, None
        for line in text.split(b"\n"):
            if line and not line.startswith(b"#") and b":" in line and b"\n" not in line:
                motif, row = line.split(b":", 1)
                vals = [int(v) for v in row.split(b",") if v != b""]
                if motif.endswith(b"+"):
                    extend_rows = {} if extend_rows is None else extend_rows
                    extend_rows.setdefault(motif[:-1], vals)
                else:
                    open_rows.setdefault(motif, vals)
        if b"A\nB" not in text:
            twin = engine.CustomIndelModel(open_rows=open_rows, extend_rows=extend_rows, default_open=info["default_open"], default_extend=info["default_extend"], lib_path=lib_path)
            assert twin.info() == info, (text, twin.info(), info)
            again = engine.custom_penalty_vectors(twin, None, bases, off)
            assert all(np.array_equal(a, b) for a, b in zip(again, got)), text
            twin.close()
        model.close()
        n_ok += 1
    assert n_ok >= n_models and n_bad >= n_models // 2 and n_default_hits > 0
    return n_ok, n_bad


def check_custom_model_in_calls(backend, tol=0.0):
    """oct_phmm_set_custom_error_model / oct_phmm_server_set_custom_error_model: NULL-vector calls get the file model's vectors (on host threads, whatever OCT_PHMM_PENALTIES says)."""
    lib_path = build_sim() if backend == "sim" else None
    rng = np.random.default_rng(404)
    text = b"# test model\nA:40,38,30,22,14,9,5,3\nT:40,38,30,22,14,9,5,3\nN:45,40,33,25,16,11,6,4\nCG:44,40,30,20,10\nNN:46,42,36,28,20,12\nNNN:47,44,40,30\nN+:3,3,4,6\nNN+:3,4,5\n"
    model = engine.CustomIndelModel(text, lib_path=lib_path)
    g = synth.make_region(rng, 12, 5, T=50, Lh=170, B=8, flank=(20, 20), positions="none")
    for h in g["haps"][1:4]:
        a = int(rng.integers(30, 90)); h[a:a + 12] = ord("A"); h[a + 20:a + 36] = np.frombuffer(b"CG" * 8, np.uint8); h[a + 40:a + 52] = np.frombuffer(b"ACT" * 4, np.uint8)
    batch = synth.batch_from_regions([g])
    want_vec = engine.custom_penalty_vectors(model, None, batch.hap_bases, batch.hap_offsets)
    basic_vec = engine.penalty_vectors(engine.default_error_model(lib_path), batch.hap_bases, batch.hap_offsets, lib_path=lib_path)
    assert not np.array_equal(want_vec[0], basic_vec[0])                        # the file's rows, not the built-in tables
    given = synth.batch_from_regions([g])
    given.gap_open, given.gap_extend, given.snv_mask_fwd, given.snv_prior_fwd, given.snv_mask_rev, given.snv_prior_rev = want_vec
    old = os.environ.get("OCT_PHMM_PENALTIES")
    try:
        for where in ("host", "device"):
            os.environ["OCT_PHMM_PENALTIES"] = where
            eng = make_engine(backend, max_indel_error=8)
            want, _ = eng.populate(given)
            eng.set_custom_error_model(model)
            rb = eng.upload(batch.without_penalty_vectors())
            for name, gvec, wvec in zip(NAMES, rb.penalty_vectors(), want_vec):
                assert np.array_equal(gvec, wvec), (where, name)
            rb.run(); got = rb.download().copy(); rb.free()
            assert np.max(np.abs(got - want), initial=0.0) <= tol
            eng.set_error_model(engine.default_error_model(lib_path))            # ... takes the file model away again
            rb = eng.upload(batch.without_penalty_vectors())
            assert np.array_equal(rb.penalty_vectors()[0], basic_vec[0]); rb.free()
            eng.close()
    finally:
        if old is None:
            os.environ.pop("OCT_PHMM_PENALTIES", None)
        else:
            os.environ["OCT_PHMM_PENALTIES"] = old
    # a row with a negative penalty: the host entry hands it over, an upload refuses it like a caller's own vector
    neg = engine.CustomIndelModel(b"A:-3\nN:5\n", lib_path=lib_path)
    eng = make_engine(backend, max_indel_error=8)
    eng.set_custom_error_model(neg)
    _, st = eng.populate(batch.without_penalty_vectors(), raise_on_error=False)
    assert st.code == abi.EINVAL
    eng.close(); neg.close()
    # the region server
    srv = engine.Server(abi.Config.default(max_indel_error=8), lib_path=lib_path)
    srv.set_custom_error_model(model)
    model.close()                                                               # the server keeps its own reference
    ref_eng = make_engine(backend, max_indel_error=8)
    want, _ = ref_eng.populate(given); ref_eng.close()
    got, st = srv.populate(batch.without_penalty_vectors(), raise_on_error=False)
    assert st.code == abi.OK and np.max(np.abs(got - want), initial=0.0) <= tol
    srv.close()
    return got.size


def check_custom_model_mutations(lib_path=None, n=1500, seed=1):
    """Well-formed model texts with 1-3 random byte edits (substitute / insert / delete over the characters a model file is made of, blanks, '\\r' and junk): the product's reader and the
    reference's make_penalty_map accept and refuse the same ones, and the accepted ones give the same gap vectors (20,000 edits by hand at the end of round 5: 5,750 accepted, 14,250 refused, by both)."""
    rng = np.random.default_rng(seed)
    alphabet = b"ACGTN:+,\n#0123456789- \r+x."
    seqs = [b"ACGT", b"AAAAAACGCGCGCGTTTACGACGACGACGTTTTTTTTTTTGAGAGAGAGAGAACCCCCCCCCAAAACAAAACAAAACAAAAC"]
    bases = np.frombuffer(b"".join(seqs), np.uint8); off = np.asarray([0, len(seqs[0]), len(seqs[0]) + len(seqs[1])], np.uint32)
    ok = bad = 0
    for _ in range(n):
        t = bytearray(random_model_text(rng, True))
        for _ in range(int(rng.integers(1, 4))):
            op, pos, c = int(rng.integers(0, 3)), int(rng.integers(0, len(t) + 1)), alphabet[int(rng.integers(0, len(alphabet)))]
            if op == 0 and pos < len(t):
                t[pos] = c
            elif op == 1:
                t.insert(pos, c)
            elif pos < len(t):
                del t[pos]
        t = bytes(t)
        rc = ref_custom_indel(t, seqs[0])[0]
        try:
            m = engine.CustomIndelModel(t, lib_path=lib_path)
        except engine.EngineError:
            assert rc != 0, ("the product refuses a text the reference accepts", t)
            bad += 1
            continue
        assert rc == 0, ("the product accepts a text the reference refuses", t, rc)
        got = engine.custom_penalty_vectors(m, None, bases, off)
        for i, s in enumerate(seqs):
            _, go, ge = ref_custom_indel(t, s)
            assert np.array_equal(got[0][off[i]:off[i + 1]], go) and np.array_equal(got[1][off[i]:off[i + 1]], ge), t
        m.close()
        ok += 1
    return ok, bad
