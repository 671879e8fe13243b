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


def init_process_group(backend: str | None = None, force: bool = False):
    """Initialises torch.distributed from the environment when WORLD_SIZE > 1 (or, with ``force``, at world size 1 too:
    a one-rank group still goes through the backend, which is how the RCCL leg is exercised on a one-GPU box).
    Returns the module or None."""
    rank, _, world = env_rank_world()
    if world <= 1 and not force:
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


class DangerGather:
    """The one collective of the path, with every buffer allocated once.

    Rank r contributes ``u8[cap][2][P]`` (mask row, V row per pair; ``cap`` = the largest shard, shorter shards are
    padded) and ``all_gather_into_tensor`` fills ``u8[world][cap][2][P]``.  When every rank owns ``cap`` pairs -- any
    batch divisible by the world size, e.g. BASELINE configs 4 and 5 -- the results are VIEWS of that buffer in global
    pair order; with ragged shards they are compacted into two more preallocated arrays.  Nothing is allocated per call
    (64 x 2304 x 2 B = 295 KB per rank at batch 512 over 8 GPUs)."""

    def __init__(self, n_pairs_total: int, P: int, device, dist):
        import torch
        self.dist = dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.n_total = int(n_pairs_total)
        self.P = int(P)
        self.counts = [shard_pairs(self.n_total, r, self.world)[1] for r in range(self.world)]
        self.cap = max(max(self.counts), 1)
        self.send = torch.zeros((self.cap, 2, self.P), dtype=torch.uint8, device=device)
        self.recv = torch.empty((self.world, self.cap, 2, self.P), dtype=torch.uint8, device=device)
        self.even = all(c == self.cap for c in self.counts)
        if self.even:
            # [world, cap, P] with strides (cap * 2P, 2P, 1): the first two axes merge without a copy
            self.mask_all = self.recv[:, :, 0].view(self.world * self.cap, self.P)
            self.v_all = self.recv[:, :, 1].view(self.world * self.cap, self.P)
        else:
            self.mask_all = torch.empty((self.n_total, self.P), dtype=torch.uint8, device=device)
            self.v_all = torch.empty((self.n_total, self.P), dtype=torch.uint8, device=device)
        self.calls = 0

    def __call__(self, mask, v):
        n = mask.shape[0]
        if n != self.counts[self.rank]:
            raise ValueError(f"rank {self.rank} holds {n} pairs, expected {self.counts[self.rank]}")
        if mask.shape[1] != self.P or v.shape != mask.shape:
            raise ValueError(f"danger maps must be [{n}, {self.P}] uint8, got {tuple(mask.shape)} / {tuple(v.shape)}")
        self.send[:n, 0].copy_(mask)
        self.send[:n, 1].copy_(v)
        self.dist.all_gather_into_tensor(self.recv.view(self.world * self.cap, 2, self.P), self.send)
        if not self.even:
            o = 0
            for r, c in enumerate(self.counts):
                self.mask_all[o:o + c].copy_(self.recv[r, :c, 0])
                self.v_all[o:o + c].copy_(self.recv[r, :c, 1])
                o += c
        self.calls += 1
        return self.mask_all, self.v_all


_gathers: dict = {}


def gather_danger_maps(mask, v, n_pairs_total: int, dist=None):
    """All-gathers per-rank danger maps into global pair order.

    mask, v: torch uint8 tensors [local_pairs, P] on this rank's device (CPU tensors under gloo).
    Returns (mask_all, v_all) of shape [n_pairs_total, P] on every rank.  Ranks may own different
    numbers of pairs (shard_pairs).  The buffers behind the result belong to a per-(group, batch, P,
    device) DangerGather that is created on the first call and reused: the next call with the same key
    overwrites them, and nothing is allocated per call.  Without an initialised process group the
    inputs are returned as they are; an initialised group of ONE rank still runs the collective."""
    if dist is None or not dist.is_initialized():
        return mask, v
    key = (dist.get_world_size(), dist.get_rank(), int(n_pairs_total), int(mask.shape[1]), str(mask.device))
    g = _gathers.get(key)
    if g is None or g.dist is not dist:
        g = _gathers[key] = DangerGather(n_pairs_total, mask.shape[1], mask.device, dist)
    return g(mask, v)


def reset_gathers():
    """Drops the cached DangerGather buffers (call before destroy_process_group in long-lived processes)."""
    _gathers.clear()


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
