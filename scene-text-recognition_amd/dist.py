"""Multi-GPU sharding of the hot path (one process per GPU, torch.distributed over RCCL).

Every (frame, channel, pyramid level) plane is independent from er_tree_extract through
classify (src/ER.cpp:50-60 has no cross-plane data flow), so frames are simply dealt out
to the ranks and no data-path collective exists.  The only exchange is the one the
reference performs implicitly when `er_track` reads every channel's strong/weak lists
(src/ER.cpp:63): a variable-length gather of the 48-byte candidate records.

  all_gather(count per rank)  ->  all_gather(records padded to the max count)

A single large frame (BASELINE configs[4]: one 3840x2160 frame, 3 channels x 12 levels) has no frames to deal out: there the
(channel, level) planes are dealt out whole, longest first (`shard_planes_lpt`); the level-0 planes bound the speed-up at
about total pixels / largest plane (~6x on 8 GPUs), which only spatial strips (8(f)-4) would lift.

Over xGMI this is latency-bound (a few KB per rank).  Works with backend "nccl" (= RCCL on
ROCm, device tensors) and "gloo" (CPU tensors, used by the world_size-2 tests).
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np

from .binding import CAND_DTYPE, StrErError

# (torch is imported by the functions that use torch.distributed, not here: the strips flow below runs over the C ABI's own
# communicators, and a process that loads the system's librccl AND PyTorch's bundled ROCm runtime holds two of them)


def shard_frames(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of frames for `rank`: (first, count); blocks differ by at most one."""
    base, rem = divmod(n_frames, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def gather_candidates(cands: np.ndarray, device, frame_offset: int = 0) -> np.ndarray:
    """All ranks receive every rank's candidates, ordered by rank.  `frame_offset` is added to
    the records' frame field first so frame numbers are global.  `device`: a torch.device."""
    import torch
    import torch.distributed as dist
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


def pyr_dims(w0: int, h0: int, level: int) -> Tuple[int, int]:
    """Size of pyramid level `level` (include/str_er.h: round(W * 2^(-level/2)), at least 1)."""
    s = math.pow(2.0, -0.5 * level)
    return max(1, int(math.floor(w0 * s + 0.5))), max(1, int(math.floor(h0 * s + 0.5)))


def frame_planes(w: int, h: int, n_levels: int, channel_mask: int) -> List[Tuple[int, int, int, int]]:
    """(ch, pyr, width, height) of every logical plane of one frame, in the library's plane order (level-major)."""
    out = []
    for lvl in range(n_levels):
        pw, ph = pyr_dims(w, h, lvl)
        for ch in range(6):
            if (channel_mask >> ch) & 1:
                out.append((ch, lvl, pw, ph))
    return out


def shard_planes_lpt(costs: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of whole planes to `world` ranks: planes sorted by cost (descending, ties by
    index) go one by one to the rank with the least load so far (ties: lowest rank).  Deterministic; every rank computes the
    same table.  Returns the plane indices of each rank, ascending."""
    order = sorted(range(len(costs)), key=lambda i: (-int(costs[i]), i))
    load = [0] * world
    mine: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        mine[r].append(i)
        load[r] += int(costs[i])
    return [sorted(m) for m in mine]


def detect_plane_share(erf, bgr: np.ndarray, share: Sequence[int], n_levels: int, channel_mask: int = 0x3F, stages: int = 7) -> np.ndarray:
    """Candidates of the planes `share` (indices into frame_planes()) of one frame, ordered by (plane, key); the `node`
    field carries the global plane index.  One call with a plane subset (`text_detect_planes` = str_er_detect_bgr_planes):
    channels, the pyramid (as deep as the share's deepest plane) and the trees stay on the device.  The context must have this
    very plane layout (n_pyr_levels, channel_mask)."""
    a = np.ascontiguousarray(bgr, dtype=np.uint8)
    h, w = a.shape[:2]
    planes = frame_planes(w, h, n_levels, channel_mask)
    prm = erf.params
    if prm.n_pyr_levels != n_levels or prm.channel_mask != channel_mask:
        raise ValueError("the context's n_pyr_levels / channel_mask are not the frame's plane layout")
    sel = np.zeros(len(planes), np.uint8)
    sel[list(share)] = 1
    if not sel.any():
        return np.zeros(0, CAND_DTYPE)
    res = erf.text_detect_planes(a, sel, stages)
    out = res.cands.copy()
    idx = {(ch, lvl): i for i, (ch, lvl, _, _) in enumerate(planes)}
    out["node"] = np.array([idx[(int(c["ch"]), int(c["pyr"]))] for c in out], np.int32) if len(out) else out["node"]
    return out


def detect_frame_strips(erf, comm, bgr: np.ndarray, stages: int = 7) -> np.ndarray:
    """One frame over `comm.world` ranks with its LEVEL-0 planes cut into strips (lifts the ~6x bound of whole-plane
    sharding on a 4K frame): collective over `comm` (binding.Comm: RCCL, or an in-process group of threads).

      1. every rank extracts strip `rank` of the level-0 planes of every channel (tile trees + the seams inside the strip);
      2. the blobs are all-gathered -- from the device buffer they were assembled in, into a device buffer of the communicator:
         with RCCL they never touch a host;
      3. level-0 plane k is put together by rank k mod world (str_er_strip_merge_ex, plane_select), the planes of the other pyramid
         levels are dealt out whole, longest first, to the ranks with the least merge work (str_er_detect_bgr_planes);
      4. the candidate records are gathered: every rank returns ALL candidates, ordered by (plane, key) as a single-GPU text_detect
         of the same context gives them.

    `erf` is this rank's context (any n_pyr_levels / channel_mask, equal on all ranks)."""
    a = np.ascontiguousarray(bgr, dtype=np.uint8)
    h, w = a.shape[:2]
    prm = erf.params
    rank, world = comm.rank, comm.world
    planes = frame_planes(w, h, prm.n_pyr_levels, prm.channel_mask)
    nch = sum(1 for p in planes if p[1] == 0)
    index = {(ch, lvl): i for i, (ch, lvl, _, _) in enumerate(planes)}
    # 1 + 2: strips of the level-0 planes, device to device.  A rank whose extract fails (a thresh_step without a sentinel level is
    # rejected only where a strip has rows above it, i.e. on ranks > 0; out of memory; ...) STILL joins the all-gather, with a length
    # of -1: the C ABI announces that in its header round, every rank gets an error and none is left waiting in the collective.
    local_err = None
    try:
        dptr, nbytes = erf.strip_extract_dev(a, rank, world)
    except StrErError as e:
        local_err, dptr, nbytes = e, 0, -1
    try:
        base, starts, sizes = comm.allgather_bytes((dptr, nbytes), device_in=True, device_out=True)
    except StrErError as e:
        raise (local_err or e) from (e if local_err else None)
    # 3: owners.  Level-0 planes round robin; the rest by LPT on top of the merge load (a merge is about 0.4 of a plane's work)
    own0 = [k for k in range(nch) if k % world == rank]
    load = [0.0] * world
    for k in range(nch):
        load[k % world] += 0.4 * planes[k][2] * planes[k][3] + planes[k][2] * planes[k][3] / world
    rest = [i for i in range(len(planes)) if planes[i][1] > 0]
    mine_rest = []
    for i in sorted(rest, key=lambda i: (-planes[i][2] * planes[i][3], i)):
        r = min(range(world), key=lambda q: (load[q], q))
        load[r] += planes[i][2] * planes[i][3]
        if r == rank:
            mine_rest.append(i)
    parts = []
    try:
        if own0:
            sel = np.zeros(nch, np.uint8)
            sel[own0] = 1
            res = erf.strip_merge_ex(a, [base + s for s in starts], sizes, device_blobs=True, plane_select=sel, stages=stages)
            c = res.cands.copy()
            if len(c):
                c["node"] = np.array([index[(int(x["ch"]), 0)] for x in c], np.int32)
            parts.append(c)
        if mine_rest:
            sel = np.zeros(len(planes), np.uint8)
            sel[mine_rest] = 1
            res = erf.text_detect_planes(a, sel, stages)
            c = res.cands.copy()
            if len(c):
                c["node"] = np.array([index[(int(x["ch"]), int(x["pyr"]))] for x in c], np.int32)
            parts.append(c)
    except StrErError as e:
        local_err = e
    mine = np.concatenate(parts) if parts else np.zeros(0, CAND_DTYPE)
    # 4: the candidate gather (the one exchange of the reference's data flow, src/ER.cpp:63); a rank whose merge failed joins it
    # as "cannot take part" for the same reason as above
    try:
        allc, _ = comm.gather(mine, failed=local_err is not None)
    except StrErError as e:
        raise (local_err or e) from (e if local_err else None)
    out = allc[np.lexsort((allc["key"], allc["node"]))]
    out["node"] = -1
    return out


def detect_frame_plane_sharded(erf, bgr: np.ndarray, rank: int, world: int, n_levels: int, channel_mask: int = 0x3F,
                               device=None, stages: int = 7) -> np.ndarray:
    """One frame, planes dealt out to the ranks (LPT by pixel count); every rank returns ALL candidates, ordered by
    (plane, key) as a single-GPU `text_detect` with the same pyramid gives them.  `erf`'s n_pyr_levels / channel_mask are the
    frame's.  Collective over the default process group when world > 1."""
    h, w = bgr.shape[:2]
    planes = frame_planes(w, h, n_levels, channel_mask)
    share = shard_planes_lpt([pw * ph for (_, _, pw, ph) in planes], world)[rank]
    mine = detect_plane_share(erf, bgr, share, n_levels, channel_mask, stages)
    if world > 1:
        import torch
        mine = gather_candidates(mine, device or torch.device("cpu"))
    out = mine[np.lexsort((mine["key"], mine["node"]))]
    out["node"] = -1
    return out
