#!/usr/bin/env python3
"""bench.py — one "step" = one pass of the hot path (HaplotypeLikelihoodArray::populate on the GPU: candidate
classification + scalar fast path, banded pair-HMM DP with and without traceback, flank walk, mapping-quality
epilogue) over one synthetic batch that is already resident in HBM.

Workload at N=1: BASELINE.json configs[2] — 100k Illumina-like 150 bp reads x 128 300 bp haplotypes, band 16,
int16 lanes, flank state 40/40 (the largest single-GPU configuration; configs[1], the 1k x 64 batch, is a parity
test and is additionally timed as `small_batch_ms`). N>1: one process per GPU, each with its own region of the same
shape (weak scaling, no collective: regions are independent), value = sum over ranks / max-over-ranks time.

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
# Packed-int16 / min / perm / DPP wave64 instructions issue at 4 cycles per SIMD on gfx950 (tools/valu_ubench.hip, measured):
# 256 CU x 4 SIMD x 2.4 GHz / 4 = 614 G wave-instructions/s is the issue peak the DP kernels run against.
VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 4
# VALU wave-instructions per DP iteration per wave (8 tasks at B = 16), counted in the ISA of this build
VALU_PER_ITER = {"score": 29.25, "trace": 51.75}
# Measured issue cost per wave64 instruction on one SIMD, in cycles of the nominal 2.4 GHz clock (profiles/r01_step16_valu_issue_rates.log):
# v_add_u32 / v_and / v_or / v_mov 2.6 ("fast"), everything packed, VOP3, DPP, perm, min 4.4 ("slow"); shares of fast instructions in the
# main loops of this build's ISA: score-only 45 of 117, traceback 103 of 207.
ISSUE_CYCLES = {"score": (45 * 2.6 + 72 * 4.4) / 117, "trace": (103 * 2.6 + 104 * 4.4) / 207}
PMC_SUMMARY = ROOT / "profiles" / "r01_step17_pmc_summary.json"


def algorithmic_bytes_per_task(T: int, B: int) -> int:
    """2T (bases + quals) + 5 (T + 2B - 1) (haplotype window, gap open/extend, SNV mask/prior) + 8 + 8."""
    return 2 * T + 5 * (T + 2 * B - 1) + 16


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
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="100kx128")
    ap.add_argument("--band", type=int, default=16)
    ap.add_argument("--regions", type=int, default=2000, help="--workload stream: active regions per rank per step (BASELINE configs[3] stand-in)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the 1k x 64 latency leg (for rocprof runs: keeps every k_dp launch full-size)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # OCT_BENCH_BACKEND=sim: test hook of the CPU suite (tests/test_bench_gloo.py) - the same launch, barrier and reduction logic on two gloo
    # ranks with the library's wave simulator in place of the GPU; never a measurement
    sim = os.environ.get("OCT_BENCH_BACKEND") == "sim"
    dev = "cpu" if sim else "cuda"
    dist = None
    if world > 1:
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
        eng = engine.Engine(cfg, lib_path=build_sim())
    else:
        eng = engine.Engine(cfg)                  # fails loudly if liboct_phmm.so / a gfx950 device is missing
    if args.workload == "stream":                 # configs[3]: a stream of independent active regions, region i -> rank i mod N, one flat batch per rank
        batch = synth.batch_from_regions(synth.region_stream(seed=42 + rank, n_regions=args.regions, B=B, positions="none"))
    else:
        batch = synth.config_batch(args.workload, seed=42 + rank, B=B, positions="none")   # candidate positions come from the device k-mer mapper
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
    cells, pairs = stats["band_cells"], stats["n_pairs"]
    n_tasks = stats["n_dp_score_only"] + stats["n_dp_traceback"]
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([float(cells), float(pairs)], dtype=torch.float64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        cells, pairs = float(c[0].item()), float(c[1].item())

    if rank == 0:
        per_step = elapsed / args.steps
        # Roofline leg (outside the timed region): the pipeline above overlaps launches of different slices, so per-launch
        # durations are taken from a single-slice run of the same batch, where every launch has the device to itself.
        # Dominant kernel = k_dp<B, TRACE=true, fast cost, FASTADD> (the traceback DP); HIP events on the library's stream.
        os.environ["OCT_PHMM_SLICES"] = "1"
        eng.set_timing(True)
        rb1 = eng.upload(batch)
        del os.environ["OCT_PHMM_SLICES"]
        rb1.run(); rb1.wait()
        kind_ms = {k: [0.0, 0] for k in ("score_fast", "trace_fast", "score_generic", "trace_generic")}
        for _ in range(3):
            rb1.run(); rb1.wait()
            for k, (ms, n) in rb1.kernel_time_by_kind().items():
                kind_ms[k][0] += ms; kind_ms[k][1] += n
        rb1.free()
        tr_ms, tr_n = kind_ms["trace_fast"]
        sc_ms, sc_n = kind_ms["score_fast"]
        avg_launch_s = (tr_ms / 1e3) / max(tr_n, 1)
        tasks_per_launch = stats["n_dp_traceback"] * 3 / max(tr_n, 1)
        alg_bytes = algorithmic_bytes_per_task(T, B) * tasks_per_launch
        achieved = alg_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # VALU view (what actually binds): wave-instructions the DP launches issued per second vs the 4-cycle issue peak
        groups = lambda n: n / (2 * (64 // B))
        valu_instr = (groups(stats["n_dp_score_only"]) * VALU_PER_ITER["score"] + groups(stats["n_dp_traceback"]) * VALU_PER_ITER["trace"]) * (T + B)
        dp_s_per_step = ((tr_ms + sc_ms + kind_ms["score_generic"][0] + kind_ms["trace_generic"][0]) / 1e3) / 3
        traffic = None
        issue_cycles = valu_instr * (ISSUE_CYCLES["score"] + ISSUE_CYCLES["trace"]) / 2
        valu_src = "loop-body wave-instructions (ISA count of the traceback form; an upper bound for late-start launches) x iterations"
        if PMC_SUMMARY.exists():        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
            pmc = json.loads(PMC_SUMMARY.read_text())
            pm = pmc.get(f"octphmm::k_dp<{B}, true, false, true>", {})
            if "hbm_read_bytes_corrected" in pm and "hbm_write_bytes" in pm:
                traffic = pm["hbm_read_bytes_corrected"] + pm["hbm_write_bytes"]
            ps = pmc.get(f"octphmm::k_dp<{B}, false, false, true>", {})
            if args.workload == "100kx128" and B == 16 and "SQ_INSTS_VALU" in pm and "SQ_INSTS_VALU" in ps:
                # the PMC passes ran this very workload: instructions actually issued per launch x launches per step
                valu_instr = pm["SQ_INSTS_VALU"] * tr_n / 3 + ps["SQ_INSTS_VALU"] * sc_n / 3
                issue_cycles = pm["SQ_INSTS_VALU"] * tr_n / 3 * ISSUE_CYCLES["trace"] + ps["SQ_INSTS_VALU"] * sc_n / 3 * ISSUE_CYCLES["score"]
                valu_src = f"SQ_INSTS_VALU per launch ({PMC_SUMMARY.relative_to(ROOT)}) x launches per step"
        out = {
            "metric": "pair-HMM band cell-updates/s", "value": cells / per_step / 1e9, "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": per_step * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16", "data": "synthetic" if not sim else "synthetic, SIMULATOR BACKEND (test hook, not a measurement)",
            "config": {"workload": (f"{args.workload}: Illumina-like 150 bp reads x 300 bp haplotypes per region, band {B}, "
                                    f"int16 lanes, flank 40/40, device k-mer mapping, one region per GPU") if args.workload != "stream" else
                                   (f"stream: {args.regions} synthetic active regions per GPU per step (R ~ lognormal(300, 0.8) in [20, 5000], H ~ min(200, geometric(24)), "
                                    f"Lh 300-500, T 150), band {B}, int16 lanes, flank 40/40, device k-mer mapping"), "band": B, "read_len": T, "hap_len": LH,
                       "pairs_per_step": pairs, "dp_tasks_per_step": n_tasks, "parallelism": f"regions sharded over {world} GPU(s), no collective"},
            "loglik_per_s": pairs / per_step,
            **({"regions_per_s": args.regions * world / per_step} if args.workload == "stream" else {}),
            "stats": stats,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "traffic_note": f"HBM bytes per launch of k_dp<{B},true,false,true> from {PMC_SUMMARY.relative_to(ROOT)} (FETCH_SIZE x 2 per the gfx950 "
                                         "correction + WRITE_SIZE, separate --pmc passes of this workload, single slice = two launches per step: late-start and full traceback); "
                                         "~85 % of it is the backpointer tile stream the walk kernel consumes, which SURVEY 8d's per-task figure does not count",
                         "kernel": f"k_dp<{B}, TRACE, fast cost, FASTADD> (traceback DP), single-slice run, HIP events on the library stream",
                         "avg_launch_ms": avg_launch_s * 1e3, "tasks_per_launch": tasks_per_launch,
                         "algorithmic_bytes_per_task": algorithmic_bytes_per_task(T, B),
                         "score_only_kernel_avg_launch_ms": (sc_ms / max(sc_n, 1)),
                         "valu": {"achieved_wave_instr_per_s": valu_instr / dp_s_per_step if dp_s_per_step > 0 else 0.0,
                                  "peak_wave_instr_per_s": VALU_PEAK_WAVE_INSTR,
                                  "frac": (valu_instr / dp_s_per_step / VALU_PEAK_WAVE_INSTR) if dp_s_per_step > 0 else 0.0,
                                  "issue_weighted_frac": (issue_cycles / (1024 * 2.4e9) / dp_s_per_step) if dp_s_per_step > 0 else 0.0,
                                  "note": "the DP is integer-VALU issue bound, not HBM bound (SURVEY.md 8d): " + valu_src +
                                          " / DP kernel time vs 256 CU x 4 SIMD x 2.4 GHz / 4 cycles; issue_weighted_frac prices each instruction at its "
                                          "measured issue cost (profiles/r01_step16_valu_issue_rates.log: 2.6 cycles for plain 32-bit add/and/or/mov, 4.4 for "
                                          "packed, VOP3, DPP, perm and min) = the share of DP kernel time the instruction stream itself accounts for"}},
        }
        if world == 1 and not args.no_small_batch:
            small = eng.upload(synth.config_batch("1kx64", seed=42, B=B, positions="none"))
            for _ in range(3):
                small.run(); small.wait()
            t1 = time.perf_counter()
            for _ in range(10):
                small.run(); small.wait()
            out["small_batch_ms"] = (time.perf_counter() - t1) / 10 * 1e3
            small.free()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, seed=42)
        print(json.dumps(out))
    rb.free()
    eng.close()
    if dist is not None:
        dist.barrier()                  # rank 0 reports (and runs its roofline leg) while the others wait: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
