"""bench.py's N > 1 launch path on CPU: the driver's own command line (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`)
with the library's wave simulator in place of the GPU and gloo in place of RCCL (OCT_BENCH_BACKEND=sim). Checks what the contract asks of
the line: ONE JSON line from rank 0, whole-job aggregates over both ranks, the steps / warmup it was given, weak scaling."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(n, workload=("--workload", "tiny")):
    from backends import build_sim
    build_sim()
    env = dict(os.environ, OCT_BENCH_BACKEND="sim")
    args = ["bench.py", "--gpus", str(n), "--steps", "2", "--warmup", "1", *workload, "--band", "8", "--no-cpu-baseline", "--no-small-batch"]
    if n == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + args
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_line_over_two_ranks_aggregates_the_whole_job():
    one, two = _run(1), _run(2)
    for b, n in ((one, 1), (two, 2)):
        assert b["n_gpus"] == n and b["steps"] == 2 and b["warmup"] == 1 and b["scaling"] == "weak" and b["higher_is_better"] is True
        assert b["unit"] == "GCUPS" and b["value"] > 0 and b["ms_per_step"] > 0 and b["vs_baseline"] is None
        assert set(b["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
        assert "SIMULATOR" in b["data"]
        assert b["verified_rows"] >= 1 and b["verified_max_abs_diff"] <= 1e-9       # the line carries its own check against the reference
    # every rank brings a region of the same shape (seed 42 + rank): the job's pairs double, the per-rank stats stay one region's
    assert two["config"]["pairs_per_step"] == 2 * one["config"]["pairs_per_step"]
    assert two["stats"]["n_pairs"] == one["stats"]["n_pairs"]
    assert abs(two["value"] * two["ms_per_step"] / (one["value"] * one["ms_per_step"]) - 2.0) < 0.2     # cells per step: twice one region's, up to the seeds


def test_bench_stream_mode_shards_one_fixed_stream_over_the_ranks():
    """--workload stream: ONE stream of regions, region i on rank i mod N (strong scaling): the job's work does not depend on N."""
    wl = ("--workload", "stream", "--regions", "6", "--stream-cap", "12", "3")
    one, two = _run(1, wl), _run(2, wl)
    for b in (one, two):
        assert b["scaling"] == "strong" and b["regions_per_s"] > 0 and b["verified_rows"] >= 1 and b["verified_max_abs_diff"] <= 1e-9
    assert two["config"]["pairs_per_step"] == one["config"]["pairs_per_step"]
    assert two["stats"]["n_pairs"] < one["stats"]["n_pairs"]                     # rank 0 of two holds regions 0, 2, 4 only
    assert abs(two["value"] * two["ms_per_step"] / (one["value"] * one["ms_per_step"]) - 1.0) < 1e-6


def test_bench_read_axis_split_of_one_region_is_strong_scaling():
    """--split reads (SURVEY.md 8e's fine split): ONE region for the whole job, rank r takes the reads [r R / N, (r + 1) R / N) against every haplotype; the job's pairs
    do not depend on N, every rank's rows are verified against the reference, and at N > 1 the line carries the per-rank host-side figures."""
    wl = ("--workload", "tiny", "--split", "reads")
    one, two = _run(1, wl), _run(2, wl)
    assert one["scaling"] == "weak" and two["scaling"] == "strong"                # (N = 1: nothing is split)
    assert two["config"]["pairs_per_step"] == one["config"]["pairs_per_step"]
    assert two["stats"]["n_pairs"] * 2 == one["stats"]["n_pairs"]                 # rank 0 of two holds half of the reads
    for b in (one, two):
        assert b["verified_rows"] >= 1 and b["verified_max_abs_diff"] <= 1e-9
    assert two["split"] == "reads" and len(two["rank_e2e_ms_from_host"]) == 2 and all(x > 0 for x in two["rank_e2e_ms_from_host"]) and len(two["rank_host_threads"]) == 2
    assert "rank_e2e_ms_from_host" not in one


def test_bench_defaults_to_the_stream_split_on_more_than_one_rank():
    """Without --workload, N > 1 runs BASELINE configs[3] (one stream sharded round-robin, strong scaling), N = 1 the 100k x 128 headline batch.
    (--regions / --stream-cap only shrink the stream to simulator size; the default size is 50,000 regions.)"""
    two = _run(2, ("--regions", "6", "--stream-cap", "12", "3"))
    assert two["scaling"] == "strong" and two["regions_per_step"] == 6 and two["regions_per_s"] > 0
    assert two["rank_ms_per_step"]["min"] > 0 and two["rank_ms_per_step"]["max"] >= two["rank_ms_per_step"]["min"]
    assert "stream" in two["config"]["workload"]
    src = (ROOT / "bench.py").read_text()
    assert 'args.regions = 50000 if (world > 1 and args.workload == "stream") else 2000' in src
