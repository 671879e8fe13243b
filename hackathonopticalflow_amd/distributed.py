"""Multi-GPU driver: one process per GPU, frame pairs sharded across ranks, danger maps gathered.

The reference processes one pair per loop turn with no state beyond ``prev_gray`` (DenseOF.py:520,525),
so pairs are independent units: rank r owns a contiguous range of pair indices and no data-path
collective is needed.  The only exchange is the per-frame danger map (uint8 mask + uint8 V per grid
point, a few KB per pair), gathered with one all-gather per batch over RCCL (``torch.distributed``
backend "nccl" on ROCm; "gloo" on CPU for the tests).  Flow fields stay on the GPU that made them.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

from .ofarn import PAIRS_CONSECUTIVE, PAIRS_INDEPENDENT


@dataclass(frozen=True)
class Shard:
    pair_start: int    # first global pair index owned by this rank
    pair_count: int
    frame_start: int   # first global frame index this rank must hold
    frame_count: int   # frames to hold (pairs share a frame in consecutive mode)


def shard_pairs(n_pairs: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced: the first n_pairs % world ranks get one extra pair."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(max(n_pairs, 0), world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def shard_frames(n_frames: int, pairs_mode: int, rank: int, world: int) -> Shard:
    """Which frames rank `rank` needs.  Consecutive (video) mode overlaps neighbours by one frame
    because pair i = frames (i, i+1) (DenseOF.py:525); independent mode pairs frames (2i, 2i+1)."""
    if pairs_mode == PAIRS_CONSECUTIVE:
        n_pairs = max(n_frames - 1, 0)
        ps, pc = shard_pairs(n_pairs, rank, world)
        return Shard(ps, pc, ps, pc + 1 if pc else 0)
    if pairs_mode == PAIRS_INDEPENDENT:
        if n_frames % 2:
            raise ValueError("independent pairs need an even number of frames")
        ps, pc = shard_pairs(n_frames // 2, rank, world)
        return Shard(ps, pc, 2 * ps, 2 * pc)
    raise ValueError("pairs_mode must be 0 or 1")


def env_rank_world() -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment (defaults 0, 0, 1)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def init_process_group(backend: str | None = None):
    """Initialises torch.distributed from the environment when WORLD_SIZE > 1.  Returns the module or None."""
    rank, _, world = env_rank_world()
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def gather_danger_maps(mask, v, n_pairs_total: int, dist=None):
    """All-gathers per-rank danger maps into global pair order.

    mask, v: torch uint8 tensors [local_pairs, P] on this rank's device (CPU tensors under gloo).
    Returns (mask_all, v_all) of shape [n_pairs_total, P] on every rank.  Ranks may own different
    numbers of pairs (shard_pairs); shorter shards are padded for the collective and trimmed after."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return mask, v
    world = dist.get_world_size()
    counts = [shard_pairs(n_pairs_total, r, world)[1] for r in range(world)]
    cap = max(counts)
    P = mask.shape[1]
    both = torch.zeros((cap, 2, P), dtype=torch.uint8, device=mask.device)
    n = mask.shape[0]
    if n != counts[dist.get_rank()]:
        raise ValueError(f"rank {dist.get_rank()} holds {n} pairs, expected {counts[dist.get_rank()]}")
    both[:n, 0] = mask
    both[:n, 1] = v
    out = torch.empty((world, cap, 2, P), dtype=torch.uint8, device=mask.device)
    dist.all_gather_into_tensor(out.view(world * cap, 2, P), both)
    parts_m = [out[r, :counts[r], 0] for r in range(world)]
    parts_v = [out[r, :counts[r], 1] for r in range(world)]
    return torch.cat(parts_m, 0), torch.cat(parts_v, 0)


class FakeCommunicator:
    """NumPy stand-in for the gather, for unit tests of the shard arithmetic without processes."""

    def __init__(self, world: int):
        self.world = world

    def gather(self, per_rank_masks, per_rank_vs, n_pairs_total):
        counts = [shard_pairs(n_pairs_total, r, self.world)[1] for r in range(self.world)]
        for r, (m, c) in enumerate(zip(per_rank_masks, counts)):
            if len(m) != c:
                raise ValueError(f"rank {r} holds {len(m)} pairs, expected {c}")
        return np.concatenate(per_rank_masks, 0), np.concatenate(per_rank_vs, 0)
