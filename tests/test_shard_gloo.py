"""N > 1 path on CPU: two processes (gloo), regions sharded round-robin, each rank runs its shard through the library
(simulator backend here, MI355X in production), rank 0 gathers and checks every region against the oracle."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from backends import make_engine
    from octopus_amd import shard, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    regions = synth.region_stream(seed=3, n_regions=5, B=8)
    for g in regions:                      # keep the simulator fast
        g["reads"], g["quals"], g["begin"] = g["reads"][:10, :60], g["quals"][:10, :60], np.minimum(g["begin"][:10], 100)
        g["reverse"], g["mapq"], g["haps"] = g["reverse"][:10], g["mapq"][:10], g["haps"][:3]
        g["pos"] = None
    mine = shard.assign(len(regions), world, rank)
    local = shard.populate_regions(regions, mine, lambda: make_engine("sim", max_indel_error=8))
    dist.barrier()
    merged = shard.gather_to_rank0(local, dist)
    if rank == 0:
        q.put({k: v.tolist() for k, v in merged.items()})
    dist.destroy_process_group()


def test_two_rank_region_sharding_matches_oracle():
    import torch.multiprocessing as mp
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle
    from backends import build_sim
    from octopus_amd import abi, shard, synth
    build_sim()
    assert shard.assign(5, 2, 0) == [0, 2, 4] and shard.assign(5, 2, 1) == [1, 3]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    regions = synth.region_stream(seed=3, n_regions=5, B=8)
    assert sorted(int(k) for k in merged) == [0, 1, 2, 3, 4]
    for i, g in enumerate(regions):
        g["reads"], g["quals"], g["begin"] = g["reads"][:10, :60], g["quals"][:10, :60], np.minimum(g["begin"][:10], 100)
        g["reverse"], g["mapq"], g["haps"] = g["reverse"][:10], g["mapq"][:10], g["haps"][:3]
        g["pos"] = None
        want, st, _ = oracle.populate(abi.Config.default(max_indel_error=8), synth.batch_from_regions([g]))
        assert st.code == abi.OK
        got = np.asarray(merged[i] if i in merged else merged[str(i)]).reshape(-1)
        assert np.array_equal(got, want), i
