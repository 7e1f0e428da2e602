"""Shape fuzz over MULTI-REGION batches: what HaplotypeLikelihoodArray::populate (haplotype_likelihood_array.cpp:105-199) is handed across a run, many regions at a time.

Round 4's one device fault lived exactly here - a region with eleven haplotype key classes made k_window_region lose keys and the region server fault - and no
test generated such a shape. A scenario is 1-64 regions with H in 1..400 (every count 1..48 is drawn uniformly), R in 20..5,000, haplotypes of 300..2,000 bases,
read lengths 40..250 (ragged inside a region), linked chunks (templates), flank states, NULL or given penalty vectors, hq or stress qualities, allele-tree or
random-edit haplotypes, random library switches; it goes through oct_phmm_populate (flat), the resident API, and oct_phmm_server_populate from 8 threads (one call
per region). Checked: a SAMPLE of regions against the reference's own populate (oracle/_ref/libref_array.so) or, where that is absent, the C oracle - and ALL
regions of the other entry points bit for bit against the flat run. Scenario generation and the oracle run in worker processes (tools/gpu_fuzz.py), the GPU
process only calls the library and compares."""
import copy
import threading

import numpy as np

from octopus_amd import abi, engine, synth

SWITCH_SETS = [
    {},
    {},
    {"OCT_PHMM_WINDOW_LDS": "0"},
    {"OCT_PHMM_LANE_MAPPER": "0"},
    {"OCT_PHMM_LANE_MAPPER": "1"},
    {"OCT_PHMM_REC_CHUNK": "16"},
    {"OCT_PHMM_DEDUP": "1"},
    {"OCT_PHMM_DEDUP": "0"},
    {"OCT_PHMM_DEVICE_SIZED": "0"},
    {"OCT_PHMM_DEVICE_SIZED": "1"},
    {"OCT_PHMM_DSL_MERGE_DP": "1"},
    {"OCT_PHMM_DSL_MERGE_DP": "0"},
    {"OCT_PHMM_WALK_STAGE": "0"},
    {"OCT_PHMM_WALK_STAGE": "2"},
    {"OCT_PHMM_MAP_MISMATCHES": "0"},
    {"OCT_PHMM_LANE_MAPPER": "1", "OCT_PHMM_DEDUP": "1", "OCT_PHMM_SLICES": "3"},
    {"OCT_PHMM_LATE_MIN_PAIRS": "0"},                               # late-start lists for every batch
    {"OCT_PHMM_LATE_MIN_PAIRS": "0", "OCT_PHMM_DEVICE_SIZED": "0"},
    {"OCT_PHMM_PAIRED": "1", "OCT_PHMM_DEVICE_SIZED": "0", "OCT_PHMM_PAIRED_MIN_RUN": "0"},          # round 6: window-paired task lists (k_pair_sort, PAIRED segments of k_dp)
    {"OCT_PHMM_PAIRED": "1", "OCT_PHMM_DEVICE_SIZED": "0", "OCT_PHMM_PAIRED_MIN_RUN": "0", "OCT_PHMM_LATE_MIN_PAIRS": "0", "OCT_PHMM_SLICES": "3"},
    {"OCT_PHMM_LATE_START": "0"},
    {"OCT_PHMM_SLICES": "4", "OCT_PHMM_LATE_MIN_PAIRS": "0"},
    {"OCT_PHMM_JOIN_LATE": "0", "OCT_PHMM_LATE_MIN_PAIRS": "0"},     # late-start lists in launches of their own behind the new scan
    {"OCT_PHMM_JOIN_LATE": "1", "OCT_PHMM_LATE_MIN_PAIRS": "0", "OCT_PHMM_SLICES": "2"},
    {"OCT_PHMM_REC_CHUNK": "64"},

    {"OCT_PHMM_DP_ROWS": "0"},
]


def _hap_count(rng, cap):
    h = int(rng.integers(1, 49)) if rng.random() < 0.7 else int(min(400, 49 + rng.geometric(1 / 60.0)))
    return max(1, min(h, cap))


def random_region(rng, band, pair_budget, scale):
    """One region within `pair_budget` (read, haplotype) pairs."""
    sim = scale == "sim"
    linked = rng.random() < (0.08 if not sim else 0.15)
    if linked:
        chunk = 500 if not sim else 60
        Lh = (1400 + int(rng.integers(0, 401))) if not sim else 260 + int(rng.integers(0, 60))
        H = _hap_count(rng, 10 if not sim else 3)
        n_long = int(rng.integers(3, 14 if not sim else 5))
        g = synth.make_linked_region(rng, n_long, H, Lh=Lh, chunk=chunk, B=band, flank=(int(rng.integers(0, Lh // 4)), int(rng.integers(0, Lh // 4))))
        return g
    T = int(rng.choice([40, 60, 76, 100, 125, 150, 150, 150, 151, 200, 250])) if not sim else int(rng.integers(24, 70))
    Lh = int(min(2000, T + 2 * band + 130 + (rng.integers(0, 200) if rng.random() < 0.8 else rng.integers(200, 1500)))) if not sim else max(130, T + 2 * band + int(rng.integers(10, 90)))
    Lh = max(Lh, 300) if not sim else Lh
    H = _hap_count(rng, 400 if not sim else 14)
    R = int(np.clip(rng.lognormal(np.log(300), 0.9), 20, 5000)) if not sim else int(rng.integers(3, 30))
    R = max(3 if sim else 20, min(R, max(20, pair_budget // H)))
    flank = None if rng.random() < 0.2 else (int(rng.integers(0, Lh // 2)), int(rng.integers(0, Lh // 2)))
    g = synth.make_region(rng, R, H, T=T, Lh=Lh, B=band, flank=flank, positions="none", indels_per_read=int(rng.random() < 0.15),
                          q_profile="hq" if rng.random() < 0.5 else "stress", hap_model="tree" if rng.random() < 0.5 and Lh >= 132 + 12 * int(np.ceil(np.log2(max(H, 2)))) else "edits")
    g["mapq"] = rng.integers(0, 255, R).astype(np.uint8) if rng.random() < 0.5 else g["mapq"]
    if rng.random() < 0.3:                                    # ragged reads
        g["read_len"] = rng.integers(min(T, max(band + 4, T // 3)), T + 1, R).astype(np.int64)   # (toy reads can be shorter than band + 4: then all stay whole)
    if rng.random() < 0.1:                                    # a few non-ACGT bytes: generic kernels beside fast ones in one batch
        g["reads"][rng.integers(0, R), rng.integers(0, min(T, 30))] = ord("N")
        g["haps"][int(rng.integers(0, H))][int(rng.integers(0, Lh))] = ord("N")
    if rng.random() < 0.2:                                    # templates of 1-2 consecutive reads
        rows, r = [0], 0
        while r < R:
            r += 2 if (R - r >= 2 and rng.random() < 0.5) else 1
            rows.append(r)
        g["row_off"] = np.asarray(rows, np.int64)
    return g


def make_scenario(seed, scale="gpu"):
    rng = np.random.default_rng([seed, 55])
    sim = scale == "sim"
    band = int(rng.choice([16, 16, 16, 16, 8, 32]))
    n_regions = int(rng.choice([1, 2, 3, 5, 8, 12, 16, 24, 40, 64])) if not sim else int(rng.integers(1, 7))
    total_budget = int(rng.choice([20_000, 60_000, 150_000, 400_000])) if not sim else 700
    regions, used = [], 0
    for _ in range(n_regions):
        g = random_region(rng, band, max(200 if not sim else 20, int(total_budget / n_regions * rng.uniform(0.3, 3.0))), scale)
        regions.append(g)
        rows = g["reads"].shape[0]
        used += rows * len(g["haps"])
        if used >= total_budget:
            break
    cfg = dict(max_indel_error=band)
    if rng.random() < 0.1: cfg["use_int_scores"] = 1
    if rng.random() < 0.2: cfg["use_mapping_quality"] = 0
    if rng.random() < 0.15: cfg["use_flank_state"] = 0
    if rng.random() < 0.2: cfg.update(mapping_quality_cap=int(rng.integers(20, 100)), mapping_quality_cap_trigger=int(rng.integers(10, 120)))
    if rng.random() < 0.15: cfg["max_mapping_positions"] = int(rng.integers(1, 15))
    switches = dict(SWITCH_SETS[int(rng.integers(0, len(SWITCH_SETS)))])
    apis = ["flat"]
    if rng.random() < 0.35: apis.append("server")
    if rng.random() < 0.25: apis.append("resident")
    null_vectors = rng.random() < 0.25
    k = min(len(regions), 3 if not sim else 2)
    sample = sorted(int(x) for x in rng.choice(len(regions), size=k, replace=False))
    big = int(np.argmax([len(g["haps"]) for g in regions]))      # the region with most haplotypes is always looked at
    if big not in sample: sample[-1] = big
    return dict(seed=seed, regions=regions, cfg=cfg, switches=switches, apis=apis, null_vectors=null_vectors, sample=sorted(set(sample)))


def _region_batch(g, null_vectors, model):
    """One region as its own single-region batch, with the vectors the error model gives its haplotypes where the scenario leaves them to the library."""
    b = synth.batch_from_regions([g])
    if null_vectors:
        import oracle
        vec = [oracle.penalty_vectors(model, h) for h in g["haps"]]
        b = copy.copy(b); b._keep = []
        b.gap_open, b.gap_extend, b.snv_mask_fwd, b.snv_prior_fwd, b.snv_mask_rev, b.snv_prior_rev = (np.concatenate([v[i] for v in vec]) for i in range(6))
    return b


def expected(scn, model=None):
    """Oracle results of the sampled regions: the reference's own populate where oracle/_ref is built, else the C restatement."""
    import oracle
    cfg = abi.Config.default(**scn["cfg"])
    if scn["null_vectors"] and model is None:
        model = engine.default_error_model()
    out = {}
    use_ref = oracle.have_ref_array() and cfg.max_mapping_positions == abi.Config.default().max_mapping_positions
    for i in scn["sample"]:
        b = _region_batch(scn["regions"][i], scn["null_vectors"], model)
        if use_ref:
            code, vals, _, _, _ = oracle.ref_array_populate(cfg, b, n_threads=1)
            assert code == 0
            out[i] = np.array(vals)
        else:
            vals, st, _ = oracle.populate(cfg, b, n_threads=1)
            assert st.code == abi.OK
            out[i] = np.array(vals)
    return out


def make_and_expect(args):
    seed, scale = args
    scn = make_scenario(seed, scale)
    return scn, expected(scn)


class Runner:
    """Engines and servers by configuration, kept across scenarios (a server is three handles with a gigabyte of scratch each)."""
    def __init__(self, backend):
        self.backend = backend
        self.lib_path = None
        if backend == "sim":
            from backends import build_sim
            self.lib_path = build_sim()
        self.model = engine.default_error_model(self.lib_path)
        self.engines, self.servers = {}, {}
        self.stats = dict(scenarios=0, regions=0, pairs=0, sampled_regions=0, server_calls=0, resident=0, null_vectors=0, switches={})

    def engine(self, cfg_kw):
        key = tuple(sorted(cfg_kw.items()))
        if key not in self.engines:
            if len(self.engines) >= 6:                      # (a handle keeps its traceback scratch and its block cache: hundreds of configurations would fill the device)
                k0 = next(iter(self.engines)); self.engines.pop(k0).close()
            self.engines[key] = engine.Engine(abi.Config.default(**cfg_kw), lib_path=self.lib_path)
            self.engines[key].set_error_model(self.model)
        return self.engines[key]

    def server(self, cfg_kw):
        key = tuple(sorted(cfg_kw.items()))
        if key not in self.servers:
            if len(self.servers) >= 2:                      # (each holds four handles with 4 GB of reserved scratch)
                k0 = next(iter(self.servers)); self.servers.pop(k0).close()
            self.servers[key] = engine.Server(abi.Config.default(**cfg_kw), lib_path=self.lib_path)
            self.servers[key].set_error_model(self.model)
        return self.servers[key]

    def close(self):
        for e in self.engines.values(): e.close()
        for s in self.servers.values(): s.close()
        self.engines, self.servers = {}, {}

    def run(self, scn, want, tol):
        regions = scn["regions"]
        for k, v in scn["switches"].items(): engine.test_set(k, v, self.lib_path)
        try:
            flat = synth.batch_from_regions(regions)
            if scn["null_vectors"]: flat = flat.without_penalty_vectors()
            eng = self.engine(scn["cfg"])
            got, st = eng.populate(flat)
            assert st.code == abi.OK
            offs = np.concatenate([[0], np.cumsum([(len(g["row_off"]) - 1 if g.get("row_off") is not None else g["reads"].shape[0]) * len(g["haps"]) for g in regions])])
            assert offs[-1] == got.size, (offs[-1], got.size)
            per_region = [got[offs[i]:offs[i + 1]] for i in range(len(regions))]
            for i, w in want.items():
                d = np.abs(per_region[i] - w)
                assert per_region[i].shape == w.shape and not (d > tol).any() and not np.isnan(per_region[i]).any(), \
                    ("flat", scn["seed"], i, len(regions[i]["haps"]), regions[i]["reads"].shape, scn["cfg"], scn["switches"], float(np.nanmax(d)))
            if "resident" in scn["apis"]:
                rb = eng.upload(flat); rb.run(); again = rb.download(); rb.free()
                assert np.array_equal(again, got), ("resident", scn["seed"], scn["cfg"], scn["switches"])
                self.stats["resident"] += 1
            if "server" in scn["apis"]:
                srv = self.server(scn["cfg"])
                singles = []
                for g in regions:
                    b = synth.batch_from_regions([g])
                    singles.append(b.without_penalty_vectors() if scn["null_vectors"] else b)
                outs, errs = [None] * len(regions), []
                def worker(t):
                    try:
                        for i in range(t, len(regions), 8):
                            o, s = srv.populate(singles[i]); assert s.code == abi.OK; outs[i] = o.copy()
                    except Exception as e:  # noqa: BLE001
                        errs.append(e)
                ths = [threading.Thread(target=worker, args=(t,)) for t in range(min(8, len(regions)))]
                [t.start() for t in ths]; [t.join() for t in ths]
                assert not errs, ("server", scn["seed"], errs[:1])
                for i in range(len(regions)):
                    assert np.array_equal(outs[i], per_region[i]), ("server vs flat", scn["seed"], i, len(regions[i]["haps"]), regions[i]["reads"].shape, scn["cfg"], scn["switches"])
                self.stats["server_calls"] += len(regions)
        finally:
            for k in scn["switches"]: engine.test_set(k, None, self.lib_path)
        s = self.stats
        s["scenarios"] += 1; s["regions"] += len(regions); s["pairs"] += int(flat.n_read_pairs()); s["sampled_regions"] += len(want)
        s["null_vectors"] += int(scn["null_vectors"])
        key = ",".join(f"{k[9:]}={v}" for k, v in sorted(scn["switches"].items())) or "default"
        s["switches"][key] = s["switches"].get(key, 0) + 1


def worker_pool(n_workers):
    """Processes that make scenarios and their oracle answers ahead of the caller. Spawned, not forked: the caller may hold a HIP context."""
    import multiprocessing as mp
    return mp.get_context("spawn").Pool(n_workers)


def check_shapes(backend, seeds, scale=None, tol=None, pool=None):
    """Run the scenarios of `seeds`; pool: a multiprocessing pool (made BEFORE the first HIP call) that generates scenarios and oracle answers ahead of the GPU."""
    scale = scale or ("sim" if backend == "sim" else "gpu")
    tol = (0.0 if backend == "sim" else 1e-9) if tol is None else tol
    r = Runner(backend)
    try:
        items = pool.imap(make_and_expect, [(s, scale) for s in seeds], chunksize=1) if pool is not None else (make_and_expect((s, scale)) for s in seeds)
        for scn, want in items:
            try:
                r.run(scn, want, tol)
            except Exception as e:
                raise AssertionError(f"scenario seed {scn['seed']} ({len(scn['regions'])} regions, cfg {scn['cfg']}, switches {scn['switches']}, apis {scn['apis']}): {e!r}") from e
    finally:
        r.close()
    return r.stats
