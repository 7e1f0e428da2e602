"""Multi-GPU sharding of the hot path: independent regions (populate() calls) go round-robin to ranks, one process per
GPU, NO collective on the data path. torch.distributed is used only to launch/barrier and, in tests, to
gather the per-region result matrices on rank 0 (control plane)."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np

from . import abi, synth


def assign(n_regions: int, world: int, rank: int) -> List[int]:
    """Region i -> device i mod G (BASELINE.json configs[3])."""
    return list(range(rank, n_regions, world))


def populate_regions(regions: List[dict], mine: List[int], make_engine: Callable[[], "object"]) -> Dict[int, np.ndarray]:
    """Score this rank's regions in ONE multi-region batch; returns region index -> H x R matrix of ln-likelihoods."""
    if not mine:
        return {}
    eng = make_engine()
    batch = synth.batch_from_regions([regions[i] for i in mine])
    out, _ = eng.populate(batch)
    res, off = {}, 0
    for i in mine:
        H, R = len(regions[i]["haps"]), regions[i]["reads"].shape[0]
        res[i] = out[off:off + H * R].reshape(H, R).copy()
        off += H * R
    eng.close()
    return res


def gather_to_rank0(local: Dict[int, np.ndarray], dist=None) -> Optional[Dict[int, np.ndarray]]:
    if dist is None or dist.get_world_size() == 1:
        return local
    bucket = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(local, bucket, dst=0)
    if dist.get_rank() != 0:
        return None
    merged: Dict[int, np.ndarray] = {}
    for d in bucket:
        merged.update(d)
    return merged
