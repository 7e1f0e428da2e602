#!/usr/bin/env python3
"""bench.py — one "step" = one pass of the hot path (HaplotypeLikelihoodArray::populate on the GPU: candidate mapping,
classification + scalar fast path, banded pair-HMM DP with and without traceback, flank walk, mapping-quality epilogue) over
one synthetic batch that is already resident in HBM.

Workload at N=1: BASELINE.json configs[2] — 100k Illumina-like 150 bp reads x 128 300 bp haplotypes, band 16, int16 lanes,
flank state 40/40 (the largest single-GPU short-read configuration). N>1: BASELINE configs[3], the split north_star names — ONE fixed
stream of --regions synthetic active regions (default 50,000), region i on rank i mod N, one process per GPU, no
collective (strong scaling; value = the whole stream's cells / max-over-ranks time). `--workload 100kx128` at N>1 keeps the weak-scaling
mode (every rank its own 100k x 128 region); `--workload stream` at N=1 runs the stream on one GPU.

After the timed region rank 0 (i) verifies the matrix the timed loop produced against the reference's own populate on a 5 % sample
of its rows, (ii) re-runs the batch single-slice with HIP events for the roofline block, (iii) at N=1 adds the PCIe-inclusive
populate, the configs[3] stream, the configs[4] long-read batch, the 1k x 64 latency case and the CPU baseline.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
# VALU issue peak per MI355X_MICROARCH.md (SIMD-32: a wave64 VALU instruction issues in 2 cycles): 256 CU x 4 SIMD x 2.4 GHz / 2
VALU_PEAK_2CYCLE = 256 * 4 * 2.4e9 / 2
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9
# Measured issue cost per wave64 instruction on one SIMD in SHADER cycles (tools/valu_ubench.hip with one 1024-thread workgroup pinned per CU, i.e.
# exactly four waves per SIMD for the whole kernel; s_memtime against the wall clock; profiles/r03_valu_issue_rates.log): plain VOP2 adds / logic / moves
# 2.40 ("cheap"), everything packed, VOP3, DPP, perm, min / max, shifts 4.25 ("dear"). (Rounds 1-2 quoted 2.6 / 4.4 "nominal 2.4 GHz cycles" from a
# grid that was not spread evenly over the CUs.)
CHEAP_CYCLES, DEAR_CYCLES = 2.40, 4.25
# Fallback instruction budget (main-loop ISA count of the shipped build, VALU wave-instructions per DP iteration per wave = 8 tasks at
# B = 16, and the share of cheap ones) for when no committed counter summary matches the kernels being timed.
# Round 6: the 100k x 128 step runs the window-PAIRED loops - score-only 104 VALU per four iterations (64 dear + 40 cheap), traceback 198 (92 + 106); the general
# loops (batches that do not pair: the streams, region-sized calls) are 117 (72 + 45) and 207 (96 + 111).
VALU_PER_ITER = {"score": 26.0, "trace": 49.5}
CHEAP_SHARE = {"score": 40 / 104, "trace": 106 / 198}


def issue_cycles_per_instr(kind: str) -> float:
    return CHEAP_SHARE[kind] * CHEAP_CYCLES + (1 - CHEAP_SHARE[kind]) * DEAR_CYCLES


def algorithmic_bytes_per_task(T: int, B: int) -> int:
    """2T (bases + quals) + 5 (T + 2B - 1) (haplotype window, gap open/extend, SNV mask/prior) + 8 + 8."""
    return 2 * T + 5 * (T + 2 * B - 1) + 16


def latest_pmc_summary():
    """The newest committed counter summary (tools/summarize_pmc.py) and whether it was collected on the kernels of this tree."""
    from octopus_amd import engine
    best = None
    for f in sorted((ROOT / "profiles").glob("r*_pmc_summary.json")):
        try:
            d = json.loads(f.read_text())
        except ValueError:
            continue
        if "_meta" in d:
            best = (f, d)
    if best is None:
        return None, None, False
    f, d = best
    return f, d, d["_meta"].get("kernel_source_sha") == engine.kernel_source_sha()


# ------------------------------------------------------------------------------------------------------------------
# checker + CPU baseline: the only places that touch oracle/ (never inside a timed GPU region)
# ------------------------------------------------------------------------------------------------------------------
def verify_against_reference(got: np.ndarray, regions, B: int, frac: float = 0.05, seed: int = 1, cfg=None):
    """Compare the matrix the timed loop left on the device with the reference's OWN HaplotypeLikelihoodArray::populate
    (oracle/_ref/libref_array*.so; the oracle restatement where that build is absent) on a sample of rows: for a one-region batch
    a random `frac` of its reads against all haplotypes, for a multi-region batch every 1/frac-th region in full. A (read, haplotype)
    result does not depend on the other reads of its call, so the sub-call reproduces the sampled rows exactly."""
    import oracle
    from octopus_amd import abi, synth
    cfg = cfg if cfg is not None else abi.Config.default(max_indel_error=B)
    cores = oracle.host_cores()
    use_ref = oracle.have_ref_array()
    n_rows = lambda g: g["reads"].shape[0] if g.get("row_off") is None else len(g["row_off"]) - 1       # likelihood rows: reads, or templates of linked reads
    isa = "avx2" if use_ref and oracle.have_ref_array("avx2") else "sse2"

    def reference(batch):
        if use_ref:
            code, want, _, _, _ = oracle.ref_array_populate(cfg, batch, n_threads=cores, isa=isa)
            assert code == 0
            return want
        return oracle.populate(cfg, batch, n_threads=cores)[0]

    rng = np.random.default_rng(seed)
    rows, worst = 0, 0.0
    if len(regions) == 1:
        g = regions[0]
        R, H = g["reads"].shape[0], len(g["haps"])
        idx = np.sort(rng.choice(R, size=max(1, int(np.ceil(R * frac))), replace=False))
        want = reference(synth.batch_from_regions([synth.subset_reads(g, idx)])).reshape(H, len(idx))
        mine = got.reshape(H, R)[:, idx]
        worst = float(np.max(np.abs(mine - want)))
        rows = len(idx)
    else:
        step = max(1, int(round(1 / frac)))
        first = int(rng.integers(0, min(step, len(regions))))
        off = np.concatenate([[0], np.cumsum([n_rows(g) * len(g["haps"]) for g in regions])])
        for i in range(first, len(regions), step):
            want = reference(synth.batch_from_regions([regions[i]]))
            worst = max(worst, float(np.max(np.abs(got[off[i]:off[i + 1]] - want), initial=0.0)))
            rows += n_rows(regions[i])
    return {"verified_rows": rows, "verified_max_abs_diff": worst,
            "verified_against": (f"the reference's own HaplotypeLikelihoodArray::populate ({isa.upper()} kernels, {cores} host threads)" if use_ref
                                 else "CPU oracle restatement (no reference build on this box)")}


def host_topology():
    """Sockets, cores and hardware threads of the HOST (from /proc/cpuinfo), beside what the cgroup lets this process use: `cpu_baseline.cores` is the latter, and
    the north star's ">= 50 x a single socket" can only be argued from a stated per-core rate times a stated core count."""
    import os
    phys, cores_of, n_threads, model = set(), {}, 0, ""
    try:
        cur = {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = (x.strip() for x in line.split(":", 1)); cur[k] = v
            elif cur:
                n_threads += 1; pid = cur.get("physical id", "0"); phys.add(pid); cores_of.setdefault(pid, set()).add(cur.get("core id", str(n_threads)))
                model = cur.get("model name", model); cur = {}
        if cur:
            n_threads += 1; pid = cur.get("physical id", "0"); phys.add(pid); cores_of.setdefault(pid, set()).add(cur.get("core id", str(n_threads)))
    except OSError:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    cps = max((len(c) for c in cores_of.values()), default=0)
    return {"host_sockets": len(phys), "host_cores_per_socket": cps, "host_threads": n_threads, "host_cpu_model": model,
            "affinity_threads": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": quota}


def cpu_baseline(B: int, seed: int, seconds_budget: float = 20.0):
    """The reference's CPU path on the host cores, on a bounded sample of the same workload. Preferred (kind "reference"): the reference's OWN
    HaplotypeLikelihoodArray::populate (haplotype_likelihood_array.cpp + model + pair_hmm.hpp + its SSE2 kernels, built in place into
    oracle/_ref/libref_array.so) with its own thread pool fanning out over haplotypes. Also reported: the oracle's restated upper layers over
    the reference's SIMD kernels (leaner than the reference's own per-call allocations), and the scalar port where no reference build exists."""
    import oracle
    from octopus_amd import abi, synth
    cores = oracle.host_cores()
    cfg = abi.Config.default(max_indel_error=B)
    rng = np.random.default_rng(seed)
    R, H = 20_000, 64          # ~10 s of single-core work per pass, spread over the host threads the cgroup grants
    batch = synth.batch_from_regions([synth.make_region(rng, R, H, B=B, positions="none")])

    def timed(fn):
        t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
        reps = 1
        while dt * reps < 4.0 and reps < 32:      # repeat the sample until the clock is meaningful, bounded
            reps *= 2
        if reps > 1:
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            dt = (time.perf_counter() - t0) / reps
        return dt, reps

    kind, backend = "port", "oracle"
    if oracle.have_ref():
        kind, backend = "reference", ("sse2" if oracle.ref_isa_supported("sse2") else "native")
    oracle.set_l1_backend(backend)
    _, st, stats = oracle.populate(cfg, batch, n_threads=cores)
    dt_layers, reps = timed(lambda: oracle.populate(cfg, batch, n_threads=cores))
    cells, pairs = stats["band_cells"], stats["n_pairs"]
    out = {"value": cells / dt_layers / 1e9, "unit": "GCUPS", "cores": cores, "kind": kind,
           "sample": f"{R} reads x {H} haplotypes of the same generator, {reps} repetition(s), L1 = reference {backend.upper()} kernels under the oracle's upper layers"
                     if kind == "reference" else f"{R} reads x {H} haplotypes, scalar C port",
           "loglik_per_s": pairs / dt_layers}
    if kind == "reference" and oracle.ref_isa_supported("avx2"):
        oracle.set_l1_backend("native")
        dt2, _ = timed(lambda: oracle.populate(cfg, batch, n_threads=cores))
        out["value_native_isa"] = cells / dt2 / 1e9
    oracle.set_l1_backend("oracle")
    if oracle.have_ref_array():
        # the reference's own populate(): one call, its ThreadPool of `cores` workers (array.cpp:167-184); the clock covers populate() only
        def time_array(isa):
            secs = oracle.ref_array_time_populate(cfg, batch, cores, 1, isa=isa)
            reps_a = 1
            while 0 < secs * reps_a < 4.0 and reps_a < 32:
                reps_a *= 2
            if reps_a > 1:
                secs = oracle.ref_array_time_populate(cfg, batch, cores, reps_a, isa=isa) / reps_a
            return secs, reps_a
        secs, reps_a = time_array("sse2")
        isa_used = "SSE2"
        if secs > 0:
            out["value_oracle_layers_over_reference_kernels"] = out["value"]
            if oracle.have_ref_array("avx2"):             # what a -march=native build of the reference runs for B = 16 x int16
                secs2, reps2 = time_array("avx2")
                if secs2 > 0:
                    out["value_sse2_build"] = cells / secs / 1e9
                    secs, reps_a, isa_used = secs2, reps2, "AVX2"
            out["value"] = cells / secs / 1e9
            out["loglik_per_s"] = pairs / secs
            out["sample"] = (f"{R} reads x {H} haplotypes of the same generator, {reps_a} repetition(s) of the reference's own "
                             f"HaplotypeLikelihoodArray::populate (built in place, {isa_used} kernels, its thread pool of {cores} workers over haplotypes)")
            # how that scales with threads on this host: the reader multiplies the per-thread rate by a socket's cores, we do not (vs_baseline stays as measured)
            isa_key = isa_used.lower()
            curve = {}
            for n in (1, 2, 4, 8, 16):
                if n > cores:
                    break
                if n == cores:
                    curve[str(n)] = out["value"]; continue
                secs_n = oracle.ref_array_time_populate(cfg, batch, n, 1, isa=isa_key)
                if secs_n > 0:
                    curve[str(n)] = cells / secs_n / 1e9
            out["scaling"] = {"unit": "GCUPS by threads of the reference's populate thread pool", "by_threads": curve,
                              "note": ("populate() fans haplotypes out over its pool only when the pool has MORE THAN TWO workers with more than one idle "
                                       "(haplotype_likelihood_array.cpp:167: workers->size() > 2 && workers->n_idle() > 1), so 1 and 2 'threads' both run the serial loop "
                                       "(the bridge builds no pool at all for n <= 2, oracle/ref_array_bridge.cpp); from 3 on the n pool workers take a haplotype each AND the "
                                       "calling thread runs the last one itself (array.cpp:177-180): n + 1 threads compute, which is why 4 'threads' give ~4.5 x one")}
            if "1" in curve:
                out["per_thread_gcups"] = curve["1"]
            # the per-thread rate of the parallel regime: the slope between the two largest pool sizes measured (the steepest linear part; the calling thread's share cancels)
            ns = sorted(int(k) for k in curve if int(k) >= 4)
            if len(ns) >= 2:
                out["per_thread_gcups_parallel_slope"] = (curve[str(ns[-1])] - curve[str(ns[-2])]) / (ns[-1] - ns[-2])
    if kind == "reference":
        out["reference_build_note"] = ("oracle/_ref = the reference's own sources compiled where they lie (oracle/Makefile), on few-line stand-ins for the headers this image lacks "
                                       "(boost::alignment allocator, the generated system.hpp, Boost optional / variant / lexical_cast, Haplotype / AlignedRead value types: oracle/ref_shim, "
                                       "no arithmetic of the path in any of them) - DESIGN.md section 7")
    out.update(host_topology())
    if out.get("host_cores_per_socket") and out.get("per_thread_gcups"):
        cps = out["host_cores_per_socket"]
        # two projections of ONE socket, both stated as projections: the single-thread rate x cores (an upper bound: no memory or pool contention), and the measured
        # rate of the lease's threads scaled to the socket's cores (what this very code did per thread under load, x cores)
        proj_ideal = out["per_thread_gcups"] * cps
        proj_measured = out["value"] / max(cores, 1) * cps
        avx512_factor = (out["value_native_isa"] / out["value_oracle_layers_over_reference_kernels"]
                         if out.get("value_native_isa") and out.get("value_oracle_layers_over_reference_kernels") else None)
        out["socket_projection"] = {"cores": cps, "gcups_single_thread_rate_x_cores": proj_ideal, "gcups_measured_per_thread_x_cores": proj_measured,
                                    "avx512_over_sse2_kernel_factor_under_the_oracle_layers": avx512_factor,
                                    "note": "projections, not measurements: the lease grants %d threads of a %d x %d-core host" % (cores, out.get("host_sockets") or 0, cps)}
        out["socket_projection_note"] = (f"one socket of this host = {cps} cores; at perfect scaling of the single-thread rate that would be "
                                         f"{proj_ideal:.1f} GCUPS - a projection, not a measurement (the lease grants {cores} threads)")
    return out


# ------------------------------------------------------------------------------------------------------------------
def stream_from_host(eng, cfg, batch, resident, n_regions, n_batches: int = 36, in_flight: int = 2, pipelined: bool = True):
    """What a caller gets who hands over HOST buffers: (i) one oct_phmm_populate of the many-region batch, PCIe both ways; (ii) the same batch
    over and over from `in_flight` host threads with a handle each - while one handle's batch computes, the others' next batches are validated, packed,
    copied up and their results stream back - the sustained rate over n_batches consecutive batches (36: the first upload and the last run of the pipeline, which
    overlap with nothing, are 5 % of a 12-batch measurement and under 2 % of this one). Arrays and `out` live in page-locked memory
    (oct_phmm_host_alloc: the DMA engines read and write the caller's buffers themselves); the `_pageable` figures are the same calls on plain numpy arrays,
    which the library stages through its own pinned buffers. Every result is compared with the resident run's matrix (which is verified against the
    reference's own populate on a sample of regions)."""
    import threading
    from octopus_amd import engine
    pool = engine.PinnedPool()
    res = {}
    same = True
    try:
        locked = pool.batch(batch)
        for tag, bt, mk in (("", locked, lambda: pool.empty(batch.out_size(), np.float64)), ("_pageable", batch, lambda: np.empty(batch.out_size()))):
            out = mk()
            eng.populate(bt, out=out)
            t0 = time.perf_counter()
            for _ in range(3):
                eng.populate(bt, out=out)
            one = (time.perf_counter() - t0) / 3
            same = same and bool(np.array_equal(out, resident))
            res.update({f"e2e_ms_from_host{tag}": one * 1e3, f"e2e_regions_per_s{tag}": n_regions / one})
            if not pipelined:
                continue
            for k in sorted({in_flight, 3}) if not tag else [in_flight]:
                engs = [eng] + [engine.Engine(cfg) for _ in range(k - 1)]
                outs = [out] + [mk() for _ in range(k - 1)]
                errors = []
                for e, o in zip(engs[1:], outs[1:]):
                    e.populate(bt, out=o)                        # warm the other handles' pools
                done = [0] * k

                def work(t):
                    try:
                        for _ in range(n_batches // k):
                            engs[t].populate(bt, out=outs[t]); done[t] += 1
                    except Exception as e:      # noqa: BLE001 - reported below: a thread's exception is otherwise lost and the rate silently wrong
                        errors.append(repr(e))
                ths = [threading.Thread(target=work, args=(t,)) for t in range(k)]
                t0 = time.perf_counter()
                [t.start() for t in ths]; [t.join() for t in ths]
                dt = time.perf_counter() - t0
                same = same and all(bool(np.array_equal(o, resident)) for o in outs)
                for e in engs[1:]:
                    e.close()
                n = sum(done)
                name = ("e2e_pipelined" if k == in_flight else f"e2e_pipelined_{k}_in_flight") + tag
                res.update({f"{name}_regions_per_s": n * n_regions / dt, f"{name}_ms_per_batch": dt / max(n, 1) * 1e3, f"{name}_batches": n})
                if errors:
                    res[f"{name}_errors"] = errors
    finally:
        pool.close()
    res["e2e_pipelined_how"] = (f"{in_flight} host threads, one handle each, oct_phmm_populate from page-locked host buffers back to back ({in_flight} batches in flight); "
                                "_pageable: the same from plain (pageable) arrays")
    res["e2e_results_equal_resident_run"] = same
    return res


def region_call_legs(stream_regions=None, stream_resident=None, B: int = 16):
    """The reference's calling pattern (one populate per active region from each region-task thread, caller.cpp:1159-1196) through the C ABI, without an
    interpreter in the way: tools/region_calls_bench (built by __graft_entry__.build()) issues one call per region (i) from one thread on one handle, (ii) from
    N threads through the region server - on 3,000 regions of 300 reads x 24 haplotypes of its own generator (continuity with round 3; answers compared with
    plain calls), and on the configs[3] stream's own regions (the very regions the `stream` leg runs as one flat batch): every answer of the
    server compared with the resident run's matrix, and a 5 % sample of the regions with the reference's own populate."""
    import subprocess
    import tempfile
    from octopus_amd import synth
    exe = ROOT / "tools" / "region_calls_bench"
    if not exe.exists():
        return {"region_calls": {"error": "tools/region_calls_bench is not built (python -c 'import __graft_entry__ as g; g.build()')"}}

    def run(args):
        r = subprocess.run([str(exe)] + args, capture_output=True, text=True, timeout=420)
        rows = []
        for line in r.stdout.splitlines():
            try:
                rows.append(json.loads(line))
            except ValueError:
                pass
        return r.returncode, rows
    pick = lambda rows, mode, th: next((x for x in rows if x.get("mode") == mode and x.get("threads") == th), {})
    try:
        rc, rows = run(["3000", "300", "24", "1", "16"])
    except Exception as e:      # noqa: BLE001
        return {"region_calls": {"error": repr(e)}}
    check = next((x for x in rows if x.get("mode") == "server vs plain calls"), {})
    one, srv16, srv1, h16 = pick(rows, "handle per thread", 1), pick(rows, "server", 16), pick(rows, "server", 1), pick(rows, "handle per thread", 16)
    legs = {"region_call_ms": one.get("ms_per_call"), "region_server_regions_per_s": srv16.get("regions_per_s"),
            "region_calls": {"regions": "3,000 synthetic active regions of 300 reads x 24 haplotypes (150 bp x 300 bp, flank 40/40), one oct_phmm call per region from host buffers",
                             "one_thread_one_handle": one, "server_1_caller": srv1, "server_16_callers": srv16, "handle_per_thread_16": h16,
                             "server_answers_equal_plain_calls": check, "rc": rc}}
    if stream_regions is not None:
        callers = [1, 16, 32, 64, 128]
        with tempfile.TemporaryDirectory() as tmp:
            synth.write_regions_file(os.path.join(tmp, "regions.bin"), stream_regions)
            try:
                rc, rows = run(["--file", os.path.join(tmp, "regions.bin"), "--out", os.path.join(tmp, "out.bin")] + [str(c) for c in callers])
                got = np.fromfile(os.path.join(tmp, "out.bin"), dtype=np.float64)
            except Exception as e:      # noqa: BLE001
                legs["region_calls"]["stream_regions_error"] = repr(e)
                return legs
        st = {"regions": f"the {len(stream_regions)} regions of the configs[3] stream (SURVEY 8d config 4: R ~ lognormal(300, 0.8) in [20, 5000], H ~ min(200, geometric(24)), Lh 300-500), "
                         "one oct_phmm call per region from host buffers", "rc": rc,
              "one_thread_one_handle": pick(rows, "handle per thread", 1),
              "server_answers_equal_plain_calls": next((x for x in rows if x.get("mode") == "server vs plain calls"), {})}
        for c in callers:
            st[f"server_{c}_callers"] = pick(rows, "server", c)
        if stream_resident is not None:
            st["answers_equal_the_flat_batch_run"] = bool(got.shape == stream_resident.shape and np.array_equal(got, stream_resident))
        if got.size == sum(g["reads"].shape[0] * len(g["haps"]) for g in stream_regions):
            v = verify_against_reference(got, stream_regions, B, frac=0.05, seed=2)
            st.update({"verified_rows": v["verified_rows"], "verified_max_abs_diff": v["verified_max_abs_diff"], "verified_against": v["verified_against"]})
        legs["region_calls"]["stream_regions"] = st
        legs["region_calls_stream_regions_per_s_64_callers"] = st.get("server_64_callers", {}).get("regions_per_s")
    return legs


def timed_resident(rb, steps: int, warmup: int = 1):
    for _ in range(warmup):
        rb.run(); rb.wait()
    t0 = time.perf_counter()
    for _ in range(steps):
        rb.run(); rb.wait()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, help="100kx128 (default at --gpus 1) or stream (default at --gpus N > 1)")
    ap.add_argument("--band", type=int, default=16)
    ap.add_argument("--regions", type=int, default=None, help="--workload stream: regions of the ONE stream that is sharded over the ranks (BASELINE configs[3]; "
                                                              "default 50000 at N > 1 = SURVEY.md 8d config 4, 2000 on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the 1k x 64 latency leg (for rocprof runs: keeps every k_dp launch full-size)")
    ap.add_argument("--stream-cap", type=int, nargs=2, default=None, metavar=("READS", "HAPS"), help="test hook: cap every stream region's size (simulator runs)")
    ap.add_argument("--no-extras", action="store_true", help="skip verification, PCIe-inclusive, stream and long-read legs (profiling runs)")
    ap.add_argument("--split", choices=("regions", "reads"), default="regions",
                    help="N > 1: 'regions' = every rank its own region(s) (default); 'reads' = SURVEY.md 8e's fine split of ONE huge region - the read axis in N contiguous "
                         "chunks, the haplotypes replicated on every rank, the matrix's rows concatenated: strong scaling of --workload 100kx128")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.workload is None:
        args.workload = "stream" if world > 1 else "100kx128"
    if args.regions is None:
        args.regions = 50000 if (world > 1 and args.workload == "stream") else 2000
    # OCT_BENCH_BACKEND=sim: test hook of the CPU suite (tests/test_bench_gloo.py) - the same launch, barrier and reduction logic on two gloo
    # ranks with the library's wave simulator in place of the GPU; never a measurement
    sim = os.environ.get("OCT_BENCH_BACKEND") == "sim"
    dev = "cpu" if sim else "cuda"
    dist = None
    if world > 1 or os.environ.get("OCT_BENCH_FORCE_DIST"):       # FORCE_DIST: exercise the RCCL init / barrier / reductions on a one-GPU box (world_size 1)
        import torch
        import torch.distributed as dist  # RCCL; only the barrier + max-reduce of the timing contract use it
        if sim:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from octopus_amd import abi, engine, synth
    B = args.band
    T, LH = 150, 300
    cfg = abi.Config.default(max_indel_error=B, device_id=0 if sim else local_rank)
    if sim:
        sys.path.insert(0, str(ROOT / "tests"))
        from backends import build_sim
        lib_path = build_sim()
        eng = engine.Engine(cfg, lib_path=lib_path)
    else:
        lib_path = None
        eng = engine.Engine(cfg)                  # fails loudly if liboct_phmm.so / a gfx950 device is missing
    stream = args.workload in ("stream", "stream-hq")
    if stream:        # configs[3]: ONE stream of independent active regions, region i -> rank i mod N, one flat batch per rank per step
        regions = synth.region_stream_shard(seed=42, n_regions=args.regions, rank=rank, world=world, B=B, positions="none", cap=args.stream_cap,
                                            workers=max(1, min(8, (os.cpu_count() or 1) // max(world, 1))), hq=args.workload == "stream-hq")
    elif args.split == "reads":     # ONE region for the whole job (seed 42 on every rank); rank r scores reads [r R / N, (r + 1) R / N) against all of its haplotypes
        whole = synth.config_region(args.workload, seed=42, B=B, positions="none")
        n_reads_whole = whole["reads"].shape[0]
        regions = [synth.subset_reads(whole, np.arange(rank * n_reads_whole // world, (rank + 1) * n_reads_whole // world))]
    else:             # candidate positions come from the device k-mer mapper
        regions = [synth.config_region(args.workload, seed=42 + rank, B=B, positions="none")]
    batch = synth.batch_from_regions(regions)
    rb = eng.upload(batch)                        # inputs resident in HBM before the timed region

    def sync_all():
        if dist is not None:
            import torch
            dist.barrier()
            if not sim:
                torch.cuda.synchronize()

    for _ in range(args.warmup):
        rb.run(); rb.wait()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rb.run(); rb.wait()
    sync_all()
    elapsed = time.perf_counter() - t0
    stats = rb.stats()
    # band cells the DP kernels actually updated: pairs whose candidates equal another pair's of the same read share its result
    # (stats *_shared, exact de-duplication) and are NOT counted here, although the reference computes them again
    cells_ref, pairs = stats["band_cells"], stats["n_pairs"]
    cells = cells_ref - stats.get("band_cells_shared", 0)
    n_score_run = stats["n_dp_score_only"] - stats.get("n_dp_score_only_shared", 0)
    n_trace_run = stats["n_dp_traceback"] - stats.get("n_dp_traceback_shared", 0)
    n_tasks = n_score_run + n_trace_run
    n_regions_all = len(regions)
    rank_ms = [elapsed / args.steps * 1e3]
    rank_e2e_ms, rank_host_threads = None, None
    if dist is not None:
        # what the first multi-GPU run will meet first is the HOST, not the kernels (N ranks pack and upload on one node's cores): every rank also times one populate
        # from its host buffers (upload + run + download, outside the timed region) and says how many hardware threads it may run on, so that a host-bound curve is recognisable
        # (on a bounded part of the rank's work - its first 2,000 regions at most: a second resident image of a 25,000-region shard, 83 GB at N = 2, would not fit beside the timed one)
        e2e_regions = regions[:2000]
        e2e_batch = batch if len(e2e_regions) == len(regions) else synth.batch_from_regions(e2e_regions)
        out_buf = np.empty(e2e_batch.out_size())
        for _ in range(2):
            eng.populate(e2e_batch, out=out_buf)
        t_e = time.perf_counter()
        for _ in range(3):
            eng.populate(e2e_batch, out=out_buf)
        my_e2e = (time.perf_counter() - t_e) / 3 * 1e3
        del out_buf, e2e_batch
    if dist is not None:
        import torch
        mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(x.item()) / args.steps * 1e3 for x in every]
        mine2 = torch.tensor([my_e2e, float(len(os.sched_getaffinity(0)))], dtype=torch.float64, device=dev)
        every2 = [torch.zeros_like(mine2) for _ in range(world)]
        dist.all_gather(every2, mine2)
        rank_e2e_ms = [float(x[0].item()) for x in every2]; rank_host_threads = [int(x[1].item()) for x in every2]
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([float(cells), float(pairs), float(n_regions_all), float(cells_ref)], dtype=torch.float64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        cells, pairs, n_regions_all, cells_ref = float(c[0].item()), float(c[1].item()), int(c[2].item()), float(c[3].item())

    if rank == 0:
        per_step = elapsed / args.steps
        extras = not args.no_extras
        verified = {}
        if extras:
            # the matrix the timed loop produced (rank 0's), against the reference's own populate on >= 5 % of its rows
            verified = verify_against_reference(rb.download(), regions, B)
        # Roofline leg (outside the timed region): the pipeline above overlaps launches of different slices, so per-launch
        # durations are taken from a single-slice run of the same batch, where every launch has the device to itself.
        # Dominant kernel = k_dp<B, TRACE=true, fast cost, FASTADD> (the traceback DP); HIP events on the library's stream.
        engine.test_set("OCT_PHMM_SLICES", "1", lib_path)
        eng.set_timing(True)
        rb1 = eng.upload(batch)
        engine.test_set("OCT_PHMM_SLICES", None, lib_path)
        rb1.run(); rb1.wait()
        kind_ms = {k: [0.0, 0] for k in ("score_fast", "trace_fast", "score_generic", "trace_generic")}
        # the shader clock while these very launches run: a one-wave probe on its own stream (oct_phmm_probe_clock), 4 ms windows, from a second thread
        clock_samples, probing = [], [True]

        def probe():
            while probing[0]:
                try:
                    clock_samples.append(eng.probe_clock(4.0))
                except Exception:      # the simulator backend has no clock
                    return
        import threading
        prober = threading.Thread(target=probe)
        if not sim:
            prober.start()
        for _ in range(3):
            rb1.run(); rb1.wait()
            for k, (ms, n) in rb1.kernel_time_by_kind().items():
                kind_ms[k][0] += ms; kind_ms[k][1] += n
        probing[0] = False
        if not sim:
            prober.join()
        measured_ghz = float(np.median(clock_samples)) if clock_samples else None
        rb1.free()
        eng.set_timing(False)
        tr_ms, tr_n = kind_ms["trace_fast"]
        sc_ms, sc_n = kind_ms["score_fast"]
        avg_launch_s = (tr_ms / 1e3) / max(tr_n, 1)
        tasks_per_launch = n_trace_run * 3 / max(tr_n, 1)
        alg_bytes = algorithmic_bytes_per_task(T, B) * tasks_per_launch
        achieved = alg_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # VALU view (what actually binds): wave-instructions the DP launches issued per second
        groups = lambda n: n / (2 * (64 // B))
        instr = {"score": groups(n_score_run) * VALU_PER_ITER["score"] * (T + B),
                 "trace": groups(n_trace_run) * VALU_PER_ITER["trace"] * (T + B)}
        valu_src = "loop-body wave-instructions (ISA count of the traceback form; an upper bound for late-start launches) x iterations"
        dp_s_per_step = ((tr_ms + sc_ms + kind_ms["score_generic"][0] + kind_ms["trace_generic"][0]) / 1e3) / 3
        traffic = None
        pmc_file, pmc, pmc_current = latest_pmc_summary()
        if pmc is not None and args.workload == "100kx128" and B == 16 and not os.environ.get("OCT_BENCH_NO_PMC_SUMMARY"):
            pm = pmc.get(f"octphmm::k_dp<{B}, true, false, true>", {})
            ps = pmc.get(f"octphmm::k_dp<{B}, false, false, true>", {})
            if "hbm_read_bytes_corrected" in pm and "hbm_write_bytes" in pm:
                traffic = pm["hbm_read_bytes_corrected"] + pm["hbm_write_bytes"]
            if "SQ_INSTS_VALU" in pm and "SQ_INSTS_VALU" in ps:
                # the counter passes ran this very workload: instructions actually issued per launch x launches per step
                instr = {"trace": pm["SQ_INSTS_VALU"] * tr_n / 3, "score": ps["SQ_INSTS_VALU"] * sc_n / 3}
                valu_src = f"SQ_INSTS_VALU per launch ({pmc_file.relative_to(ROOT)}) x launches per step"
        valu_instr = instr["score"] + instr["trace"]
        issue_cycles = instr["score"] * issue_cycles_per_instr("score") + instr["trace"] * issue_cycles_per_instr("trace")
        instr_rate = valu_instr / dp_s_per_step if dp_s_per_step > 0 else 0.0
        clock_hz = (measured_ghz or 2.4) * 1e9
        peak_at_clock = 256 * 4 * clock_hz / 2            # the 2-cycle issue peak at the clock the DP launches were measured at
        out = {
            "metric": "pair-HMM band cell-updates/s", "value": cells / per_step / 1e9, "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": per_step * 1e3,
            "higher_is_better": True, "scaling": "strong" if (stream or (args.split == "reads" and world > 1)) else "weak", "vs_baseline": None, "dtype": "int16",
            "workload": args.workload, "regions": args.regions if stream else 1,
            "data": "synthetic" if not sim else "synthetic, SIMULATOR BACKEND (test hook, not a measurement)",
            "config": {"workload": (f"{args.workload}: Illumina-like 150 bp reads x 300 bp haplotypes per region, band {B}, "
                                    f"int16 lanes, flank 40/40, device k-mer mapping, one region per GPU") if not stream else
                                   (f"stream: ONE stream of {args.regions} synthetic active regions per step (R ~ lognormal(300, 0.8) in [20, 5000], H ~ min(200, geometric(24)), "
                                    f"Lh 300-500, T 150), region i on GPU i mod {world}, band {B}, int16 lanes, flank 40/40, device k-mer mapping"), "band": B, "read_len": T, "hap_len": LH,
                       "pairs_per_step": pairs, "dp_tasks_per_step": n_tasks, "parallelism": f"regions sharded over {world} GPU(s), no collective"},
            "loglik_per_s": pairs / per_step,
            "gcups_reference_work": cells_ref / per_step / 1e9,
            "value_note": ("value counts the band cells the DP kernels updated; gcups_reference_work also counts the cells of pairs that share another "
                           "pair's result (stats *_shared: same read, candidates equal byte for byte), which the reference computes again"),
            **({"regions_per_s": n_regions_all / per_step, "regions_per_step": n_regions_all} if stream else {}),
            "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms)},
            **({"rank_e2e_ms_from_host": rank_e2e_ms, "rank_host_threads": rank_host_threads, "split": args.split,
                "rank_note": "per rank: one populate of (at most) the rank's first 2,000 regions from host buffers (upload + run + download, outside the timed region) and the hardware threads the rank may run on - "
                             "value times resident inputs only; if rank_e2e_ms_from_host grows with N while rank_ms_per_step does not, the node's host side is the bound"}
               if rank_e2e_ms is not None else {}),
            "stats": stats,
            **verified,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_ratio": (traffic / alg_bytes) if traffic and alg_bytes else None,
                         "traffic_note": (f"HBM bytes per launch of k_dp<{B},true,false,true> from {pmc_file.relative_to(ROOT)} (FETCH_SIZE x 2 per the gfx950 "
                                          "correction + WRITE_SIZE, separate --pmc passes of this workload, single slice); traffic_ratio = counter bytes / algorithmic "
                                          "bytes of the same launch") if traffic else "no committed counter summary for this workload",
                         "pmc_summary_matches_these_kernels": bool(pmc_current) if pmc is not None else None,
                         "kernel": f"k_dp<{B}, TRACE, fast cost, FASTADD> (traceback DP), single-slice run, HIP events on the library stream",
                         "avg_launch_ms": avg_launch_s * 1e3, "tasks_per_launch": tasks_per_launch,
                         "algorithmic_bytes_per_task": algorithmic_bytes_per_task(T, B),
                         "score_only_kernel_avg_launch_ms": (sc_ms / max(sc_n, 1)),
                         "valu": {"achieved_wave_instr_per_s": instr_rate,
                                  "peak_wave_instr_per_s": VALU_PEAK_2CYCLE, "frac": instr_rate / VALU_PEAK_2CYCLE,
                                  "measured_clock_ghz": measured_ghz, "clock_samples": len(clock_samples),
                                  "frac_at_measured_clock": instr_rate / peak_at_clock,
                                  "issue_weighted_frac": (issue_cycles / (256 * 4 * clock_hz) / dp_s_per_step) if dp_s_per_step > 0 else 0.0,
                                  "note": "the DP is integer-VALU issue bound, not HBM bound (SURVEY.md 8d). frac = " + valu_src + " / DP kernel time vs the "
                                          "2-cycle issue peak of MI355X_MICROARCH.md at the nominal 2.4 GHz (256 CU x 4 SIMD x 2.4 GHz / 2); frac_at_measured_clock "
                                          "uses the shader clock sampled while these launches ran (oct_phmm_probe_clock: s_memtime against the reference clock, "
                                          "one wave on its own stream). Per instruction the SIMD needs 2.40 shader cycles for plain VOP2 add / logic / mov and 4.25 "
                                          "for packed int16, min / max, perm, DPP and every VOP3 (tools/valu_ubench.hip, four waves pinned per SIMD, "
                                          "profiles/r03_valu_issue_rates.log) - not the guide's 2 -, so this mix cannot exceed ~0.55 of the 2-cycle peak; "
                                          "issue_weighted_frac prices every instruction at its measured cost = the share of DP kernel time the instruction "
                                          "stream itself accounts for (the rest: staging latency, reductions, tile flushes, tails)"}},
        }
        stream_regs_for_calls, stream_resident_for_calls = None, None
        if world == 1 and extras and not sim:
            # PCIe-inclusive: one oct_phmm_populate of the same batch from host buffers (H2D, table build, run, results streamed back)
            outbuf = np.empty(batch.out_size())
            eng.populate(batch, out=outbuf)
            t1 = time.perf_counter()
            for _ in range(3):
                eng.populate(batch, out=outbuf)
            e2e = (time.perf_counter() - t1) / 3
            out["e2e_ms_from_host"] = e2e * 1e3
            out["e2e_gcups_pcie_inclusive"] = (stats["band_cells"] - stats.get("band_cells_shared", 0)) / e2e / 1e9
            if not stream:
                # configs[3]: the 2,000-region stream in one flat batch (resident), verified like the main batch
                sregs = synth.region_stream_shard(seed=42, n_regions=args.regions, B=B, positions="none")
                sbatch = synth.batch_from_regions(sregs)
                sb = eng.upload(sbatch)
                dt = timed_resident(sb, 5)
                ss = sb.stats()
                resident = sb.download().copy()
                sv = verify_against_reference(resident, sregs, B, frac=0.05)
                sb.free()
                out["stream"] = {"ms": dt * 1e3, "regions": len(sregs), "regions_per_s": len(sregs) / dt, "gcups": (ss["band_cells"] - ss.get("band_cells_shared", 0)) / dt / 1e9,
                                 "gcups_reference_work": ss["band_cells"] / dt / 1e9, "pairs_shared": ss.get("n_pairs_shared", 0),
                                 "loglik_per_s": ss["n_pairs"] / dt, "verified_rows": sv["verified_rows"], "verified_max_abs_diff": sv["verified_max_abs_diff"]}
                out["stream"].update(stream_from_host(eng, cfg, sbatch, resident, len(sregs)))
                stream_regs_for_calls, stream_resident_for_calls = sregs, resident
                # The regime real reads live in (round-3 verdict): a current Illumina quality profile (~92 % of the bases >= Q30, ~0.35 mismatches per read) on
                # (i) the very haplotypes of the headline batch, (ii) the configs[3] stream with a caller's haplotypes (allele combinations of a few candidate
                # sites, diploid sample): the share of candidates try_naive_evaluate answers grows and the mapper + classifier become a larger part of the step.
                def leg(regs, steps, frac):
                    bt = synth.batch_from_regions(regs)
                    rb2 = eng.upload(bt)
                    dt2 = timed_resident(rb2, steps)
                    s2 = rb2.stats()
                    v2 = verify_against_reference(rb2.download(), regs, B, frac=frac)
                    rb2.free()
                    return {"ms": dt2 * 1e3, "loglik_per_s": s2["n_pairs"] / dt2, "gcups": (s2["band_cells"] - s2.get("band_cells_shared", 0)) / dt2 / 1e9,
                            "gcups_reference_work": s2["band_cells"] / dt2 / 1e9, "n_pairs": s2["n_pairs"], "n_candidates": s2["n_candidates"], "n_fast_path": s2["n_fast_path"],
                            "fast_path_share_of_candidates": s2["n_fast_path"] / max(s2["n_candidates"], 1), "dp_tasks": s2["n_dp_score_only"] + s2["n_dp_traceback"],
                            "pairs_shared": s2.get("n_pairs_shared", 0), "verified_rows": v2["verified_rows"], "verified_max_abs_diff": v2["verified_max_abs_diff"]}
                out["hq"] = dict(leg([synth.config_region("100kx128-hq", seed=42, B=B, positions="none")], 5, 0.05),
                                 workload="100kx128-hq: the headline batch's haplotypes, reads of a current Illumina quality profile (synth.Q_PROFILES['hq'])")
                out["stream_hq"] = dict(leg(synth.region_stream_shard(seed=42, n_regions=args.regions, B=B, positions="none", hq=True), 5, 0.05),
                                        workload=f"stream-hq: the {args.regions}-region stream's shapes with allele-combination haplotypes (synth.make_tree_haplotypes), diploid samples, hq reads",
                                        regions=args.regions)
                out["stream_hq"]["regions_per_s"] = args.regions / (out["stream_hq"]["ms"] / 1e3)
                # One rank's share of the 8-GPU split (configs[3] at N = 8: region i of the 50,000-region stream -> GPU i mod 8), timed on this one GPU: no
                # scaling curve can be measured here, but T(6,250 regions) against T(50,000) / 8 extrapolated from the 2,000-region figure says whether fixed
                # costs would flatten the strong-scaling line, and it runs the N > 1 default's batch size on hardware.
                def hbm_free():
                    import ctypes
                    try:
                        hip = ctypes.CDLL("libamdhip64.so")
                        free, total = ctypes.c_size_t(), ctypes.c_size_t()
                        return int(free.value) if hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0 else None
                    except OSError:
                        return None
                t1 = time.perf_counter()
                shard = synth.region_stream_shard(seed=42, n_regions=50000, rank=0, world=8, B=B, positions="none", workers=max(1, min(8, os.cpu_count() or 1)))
                t_gen = time.perf_counter() - t1
                shbatch = synth.batch_from_regions(shard)
                free0 = hbm_free()
                t1 = time.perf_counter()
                shb = eng.upload(shbatch)
                t_up = time.perf_counter() - t1
                free1 = hbm_free()
                dt = timed_resident(shb, 3)
                shs = shb.stats()
                shres = shb.download().copy()
                shv = verify_against_reference(shres, shard, B, frac=0.01)
                shb.free()
                sh_e2e = stream_from_host(eng, cfg, shbatch, shres, len(shard), pipelined=False)
                out["stream_shard_1_of_8"] = {"regions": len(shard), "of_stream": 50000, "ms": dt * 1e3, "regions_per_s": len(shard) / dt, "pairs": shs["n_pairs"],
                                              "gcups": (shs["band_cells"] - shs.get("band_cells_shared", 0)) / dt / 1e9, "loglik_per_s": shs["n_pairs"] / dt,
                                              "generate_s": t_gen, "upload_ms": t_up * 1e3, "hbm_bytes_resident": (free0 - free1) if free0 and free1 else None,
                                              "regions_per_s_of_the_2000_region_stream": out["stream"]["regions_per_s"],
                                              "e2e_ms_from_host": sh_e2e.get("e2e_ms_from_host"), "e2e_regions_per_s": sh_e2e.get("e2e_regions_per_s"), "e2e_ms_from_host_pageable": sh_e2e.get("e2e_ms_from_host_pageable"),
                                              "e2e_results_equal_resident_run": sh_e2e.get("e2e_results_equal_resident_run"),
                                              "verified_rows": shv["verified_rows"], "verified_max_abs_diff": shv["verified_max_abs_diff"],
                                              "note": "rank 0's share of `bench.py --gpus 8` (the driver's SCALE run) on one GPU; N > 1 itself stays unmeasured"}
                del shres, shbatch, shard
                # configs[4]: 64 x 10 kb reads, 8 x 20 kb haplotypes, band 256, int32 lanes (streaming DP kernels, traceback in HBM)
                lcfg = abi.Config.default(max_indel_error=256, use_int_scores=1, device_id=local_rank)
                leng = engine.Engine(lcfg)
                lregs0 = [synth.config_region("long64x8", seed=42, B=256, positions="none")]
                lb = leng.upload(synth.batch_from_regions(lregs0))
                dt = timed_resident(lb, 3)
                ls = lb.stats()
                lgot0 = lb.download().copy()
                lb.free(); leng.close()
                lv0 = verify_against_reference(lgot0, lregs0, 256, frac=0.1, cfg=lcfg)
                out["long_read"] = {"ms": dt * 1e3, "gcups": ls["band_cells"] / dt / 1e9, "dtype": "int32", "band": 256,
                                    "verified_rows": lv0["verified_rows"], "verified_max_abs_diff": lv0["verified_max_abs_diff"],
                                    "workload": "long64x8: 64 x 10 kb reads x 8 x 20 kb haplotypes (BASELINE configs[4])", "dp_tasks": ls["n_dp_score_only"] + ls["n_dp_traceback"]}
                # the same shape at throughput size (8 x the reads: several waves per SIMD instead of less than one)
                leng = engine.Engine(lcfg)
                lregs = [synth.config_region("long512x8", seed=42, B=256, positions="none")]
                lb = leng.upload(synth.batch_from_regions(lregs))
                dt = timed_resident(lb, 2)
                ls = lb.stats()
                lgot = lb.download().copy()
                lb.free(); leng.close()
                lv = verify_against_reference(lgot, lregs, 256, frac=0.03, cfg=lcfg)
                n_tasks_l = ls["n_dp_score_only"] + ls["n_dp_traceback"]
                out["long512x8"] = {"ms": dt * 1e3, "gcups": ls["band_cells"] / dt / 1e9, "dtype": "int32", "band": 256, "dp_tasks": n_tasks_l,
                                    "waves_per_simd_one_wave_per_task": n_tasks_l / 1024.0, "verified_rows": lv["verified_rows"], "verified_max_abs_diff": lv["verified_max_abs_diff"],
                                    "workload": "long512x8: 512 x 10 kb reads x 8 x 20 kb haplotypes, band 256, int32 lanes"}
                # Long reads the way the reference's own PacBio configuration hands them to the likelihood model (resources/configs/PacBioCCS.config: max-read-length=500,
                # split-long-reads, read-linkage=LINKED, max-indel-errors=16): 500-base linked chunks, one likelihood row per long read (template), band 16, int16 lanes
                cregs = synth.linked_stream(42, 1000, B=16)
                ccfg = abi.Config.default(max_indel_error=16, device_id=local_rank)
                ceng = engine.Engine(ccfg)
                cb = ceng.upload(synth.batch_from_regions(cregs))
                dt = timed_resident(cb, 3)
                cs = cb.stats()
                cgot = cb.download().copy()
                cb.free(); ceng.close()
                cv = verify_against_reference(cgot, cregs, 16, frac=0.05, cfg=ccfg)
                out["ccs_linked"] = {"ms": dt * 1e3, "gcups": (cs["band_cells"] - cs.get("band_cells_shared", 0)) / dt / 1e9, "gcups_reference_work": cs["band_cells"] / dt / 1e9,
                                     "regions": len(cregs), "regions_per_s": len(cregs) / dt, "loglik_per_s": cgot.size / dt, "chunk_pairs_per_s": cs["n_pairs"] / dt,
                                     "dtype": "int16", "band": 16, "dp_tasks": cs["n_dp_score_only"] + cs["n_dp_traceback"], "verified_rows": cv["verified_rows"],
                                     "verified_max_abs_diff": cv["verified_max_abs_diff"],
                                     "workload": "ccs-linked: 1,000 regions of ~45 long reads cut into 500-base linked chunks (PacBioCCS.config), haplotypes of 1.4-1.8 kb, band 16"}
                # ... and unsplit (what --split-long-reads=false, or the realigner's model of option_collation.cpp:1687-1713, would ask): 10-14 kb reads at band 16, int32 lanes
                kcfg = abi.Config.default(max_indel_error=16, use_int_scores=1, use_mapping_quality=0, device_id=local_rank)
                keng = engine.Engine(kcfg)
                kregs = [synth.config_region("ccs256x12", seed=42, B=16, positions="none")]
                kb = keng.upload(synth.batch_from_regions(kregs))
                dt = timed_resident(kb, 3)
                ks = kb.stats()
                kgot = kb.download().copy()
                kb.free(); keng.close()
                kv = verify_against_reference(kgot, kregs, 16, frac=0.1, cfg=kcfg)
                out["ccs256x12"] = {"ms": dt * 1e3, "gcups": ks["band_cells"] / dt / 1e9, "dtype": "int32", "band": 16, "dp_tasks": ks["n_dp_score_only"] + ks["n_dp_traceback"],
                                    "verified_rows": kv["verified_rows"], "verified_max_abs_diff": kv["verified_max_abs_diff"],
                                    "workload": "ccs256x12: 256 unsplit HiFi-like reads of 10-14 kb x 12 haplotypes of 16 kb, band 16, int32 lanes, no mapping-quality floor"}
                # the same shape with eight times the reads: enough tasks (one per 16 lanes, int32) to fill the chip - ccs256x12 is 1.3 waves per SIMD
                keng = engine.Engine(kcfg)
                kregs = [synth.config_region("ccs2048x12", seed=42, B=16, positions="none")]
                kb = keng.upload(synth.batch_from_regions(kregs))
                dt = timed_resident(kb, 2)
                ks = kb.stats()
                kgot = kb.download().copy()
                kb.free(); keng.close()
                kv = verify_against_reference(kgot, kregs, 16, frac=0.01, cfg=kcfg)
                out["ccs2048x12"] = {"ms": dt * 1e3, "gcups": ks["band_cells"] / dt / 1e9, "dtype": "int32", "band": 16, "dp_tasks": ks["n_dp_score_only"] + ks["n_dp_traceback"],
                                     "verified_rows": kv["verified_rows"], "verified_max_abs_diff": kv["verified_max_abs_diff"],
                                     "workload": "ccs2048x12: 2,048 unsplit HiFi-like reads of 10-14 kb x 12 haplotypes of 16 kb, band 16, int32 lanes"}
                # The wrapper's other long-read instantiations (simd_pair_hmm_wrapper.hpp:207-241), which still run round 1's streaming kernel k_dp_wide + the lockstep walker (DESIGN section 6:
                # only band 16 x int32 has k_dp_rows, only bands 128 / 256 x int32 have k_dp_mw): reported so that the gap is a number - unsplit 10-14 kb reads at band 32 with int32 lanes,
                # and at band 16 with int16 lanes (whose scores wrap for reads this long exactly as the reference's do: the comparison is bit for bit all the same)
                for tag, band, wide in (("ccs256x12_band32_int32", 32, 1), ("ccs256x12_band16_int16", 16, 0)):
                    wcfg = abi.Config.default(max_indel_error=band, use_int_scores=wide, use_mapping_quality=0, device_id=local_rank)
                    weng = engine.Engine(wcfg)
                    wregs = [synth.config_region("ccs256x12", seed=42, B=band, positions="none")]
                    wb = weng.upload(synth.batch_from_regions(wregs))
                    dt = timed_resident(wb, 2)
                    ws = wb.stats()
                    wgot = wb.download().copy()
                    wb.free(); weng.close()
                    wv = verify_against_reference(wgot, wregs, band, frac=0.05, cfg=wcfg)
                    out[tag] = {"ms": dt * 1e3, "gcups": ws["band_cells"] / dt / 1e9, "dtype": "int32" if wide else "int16", "band": band, "dp_tasks": ws["n_dp_score_only"] + ws["n_dp_traceback"],
                                "verified_rows": wv["verified_rows"], "verified_max_abs_diff": wv["verified_max_abs_diff"], "kernel": "k_dp_wide (streaming, round 1's form)",
                                "workload": f"ccs256x12 at band {band}, {'int32' if wide else 'int16'} lanes: 256 unsplit reads of 10-14 kb x 12 haplotypes of 16 kb"}
        if world == 1 and extras and not sim:
            out.update(region_call_legs(stream_regs_for_calls, stream_resident_for_calls, B))
        if world == 1 and not args.no_small_batch:
            small = eng.upload(synth.config_batch("1kx64", seed=42, B=B, positions="none"))
            for _ in range(3):
                small.run(); small.wait()
            t1 = time.perf_counter()
            for _ in range(50):                      # (10 repetitions scattered between 0.48 and 0.59 ms from run to run)
                small.run(); small.wait()
            out["small_batch_ms"] = (time.perf_counter() - t1) / 50 * 1e3
            small.free()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, seed=42)
            cb = out["cpu_baseline"]
            if cb["unit"] == out["unit"] and cb["value"] > 0:
                # vs_baseline stays null: BASELINE.md publishes no number for this metric. The measured GPU / CPU ratios live in cpu_baseline, beside what they are ratios OF.
                cb["gpu_over_cpu_measured"] = out["value"] / cb["value"]
                cb["gpu_over_cpu_measured_reference_work"] = out["gcups_reference_work"] / cb["value"]
                sp = cb.get("socket_projection")
                if sp:
                    f512 = sp.get("avx512_over_sse2_kernel_factor_under_the_oracle_layers") or 1.0
                    # the north star's ">= 50 x a single socket": GPU work the reference would have to do (gcups_reference_work) over the socket projections
                    cb["vs_socket_projection"] = {
                        "avx2_single_thread_rate_x_cores": out["gcups_reference_work"] / sp["gcups_single_thread_rate_x_cores"],
                        "avx2_measured_per_thread_x_cores": out["gcups_reference_work"] / sp["gcups_measured_per_thread_x_cores"],
                        "avx512_kernels_single_thread_rate_x_cores": out["gcups_reference_work"] / (sp["gcups_single_thread_rate_x_cores"] * max(f512, 1.0)),
                        "target": 50.0,
                        "note": "GPU reference-work rate / projected ONE socket of this host; projections as described in socket_projection - the clause is met only where these are >= 50"}
                cb["ratio_note"] = (f"gpu_over_cpu_measured = value / cpu_baseline.value on {cb['cores']} host threads (the cgroup's quota, not a socket). value counts the cells the GPU kernels "
                                    "updated, the CPU figure every cell the reference evaluates: ..._reference_work compares the two on the same job")
    rb.free()
    eng.close()
    if dist is not None:
        dist.barrier()                  # rank 0 reports (and runs its roofline leg) while the others wait: leave together
        dist.destroy_process_group()
    if rank == 0:
        # The ONE JSON line is the last thing on stdout: RCCL writes its version banner through C stdio, which is block-buffered when stdout is a pipe or a file and would
        # otherwise land BEHIND the line at exit (seen on hardware with WORLD_SIZE=1 + OCT_BENCH_FORCE_DIST: profiles/r06_s28_*). Flush C's buffers first, then print and flush.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
