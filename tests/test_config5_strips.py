"""BASELINE configs[4] whole -- one 3840x2160 frame, 3 channels x 12 pyramid levels = 36 planes -- on one GPU against the oracle, and
the same frame with its level-0 planes cut into strips: `dist.detect_frame_strips` with 2 and 4 ranks (threads with a
context each on this one device, exchanging over the in-process group of the C ABI) gives the fused call's records.  Plus the device-blob
path over RCCL with a world of one, strips in a pyramid context at a small size, and what a damaged blob does (STR_ER_EFORMAT)."""
import os
import subprocess
import sys
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import check_plane_against_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ["frame", "ch", "pyr", "level", "cls", "x", "y", "w", "h", "area", "key", "score_strong", "score_weak"]


def _ranks(S, comms, make_filter, frame, stages=7):
    """dist.detect_frame_strips on every rank of an in-process group, one thread and one context per rank."""
    world = len(comms)
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            f = make_filter()
            out[r] = S.dist.detect_frame_strips(f, comms[r], frame, stages)
            f.close()
        except Exception as e:                                       # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    assert not errs, errs
    return out


def test_config5_whole_frame_against_the_oracle_and_in_strips(S, cascade_paths, oracle, oracle_cascades):
    W, H, LV, MASK = 3840, 2160, 12, 0x07

    def make():
        f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=1, n_pyr_levels=LV, channel_mask=MASK))
        f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
        return f

    frame = S.synth.stext_bgr(S.synth.frame_seed(9), W, H)
    f = make()
    fused = f.text_detect(frame, want_nodes=True)
    assert len(fused.planes) == 36
    assert [(p.pyr, p.ch) for p in fused.planes] == [(l, c) for l in range(LV) for c in range(3)]
    assert sum(p.width * p.height for p in fused.planes) == 49752702           # 49.75 Mpx per config-5 frame
    six = oracle.compute_channels(frame)
    pyr = {c: oracle.pyramid(six[c], LV) for c in range(3)}

    def one(p):
        check_plane_against_oracle(oracle, p, pyr[p.ch][p.pyr], oracle_cascades)
        return p.n_pool
    with ThreadPoolExecutor(8) as ex:
        n_pool = sum(ex.map(one, fused.planes))               # ALL 36 planes, every node table, pool, class and score
    assert n_pool == len(fused.cands) > 1500
    f.close()
    # the same frame, level-0 planes in strips over 2 and 4 ranks
    for world in (2, 4):
        out = _ranks(S, S.Comm.local_group(world), make, frame)
        for r in range(world):
            assert out[r][FIELDS].tolist() == fused.cands[FIELDS].tolist(), (world, r)


def test_a_rank_that_cannot_extract_takes_every_rank_out_of_the_collective(S, cascade_paths):
    """thresh_step 3 has no sentinel level: str_er_strip_extract_dev rejects it only where a strip has rows above it, i.e. on ranks > 0,
    while rank 0 extracts fine and enters the all-gather (it then waited forever).  The failing ranks now join the exchange as
    "cannot take part", so EVERY rank raises -- within seconds -- and the failing rank reports its own error."""
    W, H = 640, 360
    frame = S.synth.stext_bgr(S.synth.frame_seed(3), W, H)
    for world in (2, 3):
        comms = S.Comm.local_group(world)
        errs, done = [None] * world, [False] * world

        def rank_main(r):
            f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=1, thresh_step=3, channel_mask=0x07))
            f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
            try:
                S.dist.detect_frame_strips(f, comms[r], frame)
            except S.StrErError as e:
                errs[r] = e
            done[r] = True
            f.close()

        th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(60)
        assert all(done), "a rank is still waiting in the collective"
        assert all(e is not None for e in errs), errs
        assert "thresh_step" in str(errs[1])            # the rank that failed says why


def test_strips_in_a_pyramid_context_and_bad_blobs(S, cascade_paths, oracle, oracle_cascades):
    """Strips are cut from the level-0 planes of a pyramid context too; device blobs and host blobs are the same bytes; a merge with
    plane_select gives that channel's records; blobs that were damaged on the way are refused, not followed."""
    W, H, LV = 448, 300, 3

    def make(mask=0x3F):
        f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=1, n_pyr_levels=LV, channel_mask=mask, kept_cap=40000, pool_cap=10000))
        f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
        return f

    rng = np.random.default_rng(5)
    for frame in (S.synth.stext_bgr(S.synth.frame_seed(41), W, H), rng.integers(0, 256, (H, W, 3), dtype=np.uint8)):
        owner = make()
        fused = owner.text_detect(frame)
        lvl0 = fused.cands[fused.cands["pyr"] == 0]
        workers = [make() for _ in range(3)]
        blobs = [workers[s].strip_extract(frame, s, 3) for s in range(3)]
        merged = owner.strip_merge(frame, blobs)
        assert merged.cands[FIELDS].tolist() == lvl0[FIELDS].tolist()
        sel = np.array([0, 1, 0, 0, 0, 1], np.uint8)
        part = owner.strip_merge_ex(frame, blobs, plane_select=sel)
        exp = lvl0[(lvl0["ch"] == 1) | (lvl0["ch"] == 5)]
        assert part.cands[FIELDS].tolist() == exp[FIELDS].tolist()
        # whole flow on 3 ranks == the fused pyramid call
        out = _ranks(S, S.Comm.local_group(3), make, frame)
        assert all(o[FIELDS].tolist() == fused.cands[FIELDS].tolist() for o in out)
        # damaged blobs: truncated, a node id out of range in the border rows, a parent id out of range in a record
        import struct
        with pytest.raises(S.StrErError):
            owner.strip_merge(frame, [blobs[0][:-4]] + blobs[1:])
        b1 = bytearray(blobs[1])
        b1[-4:] = struct.pack("<I", 0x00F00000)                     # last entry of the last plane's bottom row: far beyond its records
        with pytest.raises(S.StrErError) as e1:
            owner.strip_merge(frame, [blobs[0], bytes(b1), blobs[2]])
        assert e1.value.code == -5
        head = 48 + 24 * 6
        rec0 = (head + 255) // 256 * 256
        b0 = bytearray(blobs[0])
        n0 = struct.unpack_from("<I", b0, 48 + 4)[0]                # plane 0: n_nodes
        assert n0 > 0
        struct.pack_into("<I", b0, rec0, 0x0A000000 | 0x00FFFFF0)   # record 0: parent id far beyond the strip's records
        with pytest.raises(S.StrErError) as e2:
            owner.strip_merge(frame, [bytes(b0)] + blobs[1:])
        assert e2.value.code == -5
        # a CYCLE of parents between two records of one level (tools/san_fuzz.py found that a damaged blob could hang the merge: every walk
        # up a parent chain spins for ever): caught by the forest check, EFORMAT, and quickly
        recs = np.frombuffer(bytes(blobs[0]), np.uint32, count=8 * n0, offset=rec0).reshape(n0, 8)
        lv = recs[:, 1] >> 24
        pair = next(((i, j) for i in range(n0) for j in range(i + 1, n0) if lv[i] == lv[j]), None)
        assert pair is not None
        b0 = bytearray(blobs[0])
        struct.pack_into("<I", b0, rec0 + 32 * pair[0], (int(lv[pair[1]]) << 24) | pair[1])
        struct.pack_into("<I", b0, rec0 + 32 * pair[1], (int(lv[pair[0]]) << 24) | pair[0])
        import time as _time
        t0 = _time.time()
        with pytest.raises(S.StrErError) as e3:
            owner.strip_merge(frame, [bytes(b0)] + blobs[1:])
        assert e3.value.code == -5 and _time.time() - t0 < 20
        # ... a parent of a LOWER level than the node, and a box outside the plane
        b0 = bytearray(blobs[0])
        hi = int(np.argmax(lv)); lo = int(np.argmin(lv))
        if lv[hi] > lv[lo]:
            struct.pack_into("<I", b0, rec0 + 32 * hi, (int(lv[lo]) << 24) | lo)
            with pytest.raises(S.StrErError) as e4:
                owner.strip_merge(frame, [bytes(b0)] + blobs[1:])
            assert e4.value.code == -5
        b0 = bytearray(blobs[0])
        struct.pack_into("<I", b0, rec0 + 32 * 0 + 24, 60000)        # x1 far outside the plane
        with pytest.raises(S.StrErError) as e5:
            owner.strip_merge(frame, [bytes(b0)] + blobs[1:])
        assert e5.value.code == -5
        again = owner.strip_merge(frame, blobs)                     # ... and the context is fine afterwards
        assert again.cands[FIELDS].tolist() == lvl0[FIELDS].tolist()
        for f in workers + [owner]:
            f.close()


_RCCL_STRIPS = r"""
import sys, tempfile
sys.path.insert(0, %r)
import numpy as np
import str_er_amd as S
comm = S.Comm.rccl(0, 0, 1, S.Comm.unique_id())
sp, wp = S.cascade_io.write_golden(tempfile.mkdtemp())
erf = S.ERFilter(params=S.Params(max_width=640, max_height=480, max_frames=1, n_pyr_levels=4, channel_mask=7))
erf.load_cascade(0, sp); erf.load_cascade(1, wp)
frame = S.synth.stext_bgr(S.synth.frame_seed(2), 640, 480)
fused = erf.text_detect(frame)
fields = ["frame", "ch", "pyr", "level", "cls", "x", "y", "w", "h", "area", "key", "score_strong", "score_weak"]
# the device blob goes through ncclAllGather into the communicator's device buffer and is merged from there
dptr, n = erf.strip_extract_dev(frame, 0, 1)
host_copy = erf.strip_extract(frame, 0, 1)
base, starts, sizes = comm.allgather_bytes((dptr, n), device_in=True, device_out=True)
assert sizes == [n] and len(host_copy) == n
assert comm.allgather_bytes((dptr, n), device_in=True) == [host_copy]
out = S.dist.detect_frame_strips(erf, comm, frame)
assert out[fields].tolist() == fused.cands[fields].tolist() and len(out) > 0
comm.close()
print("rccl strips ok", len(out), n)
"""


def test_rccl_device_blobs_world_of_one():
    """The device-to-device path: blob assembled on the GPU, ncclAllGather (RCCL through dlopen) into the communicator's device
    buffer, merged from there; and the whole of dist.detect_frame_strips over an RCCL communicator.  World of one: the boxes have
    one GPU.  (In a process of its own, see test_gather_native.)"""
    r = subprocess.run([sys.executable, "-c", _RCCL_STRIPS % ROOT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl strips ok" in r.stdout, r.stdout + r.stderr
