"""Independent restarts across GPUs: restart r runs on rank
r mod world_size, there is no data-path collective, and the only exchange is a
host-side gather of one (score, solution) record per restart followed by a
minimum in ScoredSolution order (score first, then the solution vector:
local-search/src/local_search.rs:29-37).  Works on any torch.distributed backend
(RCCL under `nccl` on the GPUs, `gloo` in the CPU tests)."""
from __future__ import annotations

from typing import Iterable, Sequence


def shard_restarts(n_restarts: int, world_size: int, rank: int) -> list[int]:
    """Global restart ids owned by `rank`: r mod world_size == rank."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside [0, {world_size})")
    return list(range(rank, n_restarts, world_size))


def record(restart_id: int, score, solution: Iterable[int]) -> tuple:
    """One restart's best: orders exactly like ScoredSolution (score, solution)."""
    score_key = tuple(score) if isinstance(score, (tuple, list)) else (int(score),)
    return (tuple(int(s) for s in score_key), tuple(int(x) for x in solution), int(restart_id))


def best_of(records: Sequence[tuple]):
    """Minimum by (score, solution vector); restart id only breaks exact duplicates."""
    return min(records) if records else None


def gather_best(local_records: Sequence[tuple], group=None):
    """Every rank contributes its restarts' records; every rank gets (global best, all records)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        recs = list(local_records)
        return best_of(recs), recs
    world = dist.get_world_size(group)
    gathered = [None] * world
    dist.all_gather_object(gathered, list(local_records), group=group)
    recs = [r for part in gathered for r in part]
    return best_of(recs), recs
