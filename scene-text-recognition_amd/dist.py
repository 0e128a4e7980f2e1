"""Multi-GPU sharding of the hot path (one process per GPU, torch.distributed over RCCL).

Every (frame, channel, pyramid level) plane is independent from er_tree_extract through
classify (src/ER.cpp:50-60 has no cross-plane data flow), so frames are simply dealt out
to the ranks and no data-path collective exists.  The only exchange is the one the
reference performs implicitly when `er_track` reads every channel's strong/weak lists
(src/ER.cpp:63): a variable-length gather of the 48-byte candidate records.

  all_gather(count per rank)  ->  all_gather(records padded to the max count)

Over xGMI this is latency-bound (a few KB per rank).  Works with backend "nccl" (= RCCL on
ROCm, device tensors) and "gloo" (CPU tensors, used by the world_size-2 tests).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .binding import CAND_DTYPE


def shard_frames(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of frames for `rank`: (first, count); blocks differ by at most one."""
    base, rem = divmod(n_frames, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def gather_candidates(cands: np.ndarray, device: torch.device, frame_offset: int = 0) -> np.ndarray:
    """All ranks receive every rank's candidates, ordered by rank.  `frame_offset` is added to
    the records' frame field first so frame numbers are global."""
    assert cands.dtype == CAND_DTYPE
    world = dist.get_world_size()
    local = cands.copy()
    if frame_offset:
        local["frame"] += np.uint32(frame_offset)
    n_local = torch.tensor([len(local)], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts_h = [int(c.item()) for c in counts]
    cap = max(max(counts_h), 1)
    buf = torch.zeros(cap * CAND_DTYPE.itemsize, dtype=torch.uint8, device=device)
    if len(local):
        raw = torch.from_numpy(local.view(np.uint8).reshape(-1).copy())
        buf[: raw.numel()] = raw.to(device)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    parts: List[np.ndarray] = []
    for r in range(world):
        raw = outs[r][: counts_h[r] * CAND_DTYPE.itemsize].cpu().numpy()
        parts.append(raw.view(CAND_DTYPE).copy())
    return np.concatenate(parts) if parts else np.zeros(0, CAND_DTYPE)
