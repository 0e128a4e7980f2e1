"""The C-ABI candidate gather (gather.cpp: str_er_comm_* / str_er_gather_*).  On the CPU the packing code -- counts,
padding to the largest count, dropping the padding, frame offsets -- runs over the in-process transport with one thread per
rank (world size 2 and 3, ragged and empty shares); on the GPU box the RCCL transport runs with a world of one and takes the
records straight from the device array of the last detect call."""
import threading

import numpy as np
import pytest


def _fake_cands(S, rank, n, step):
    c = np.zeros(n, S.CAND_DTYPE)
    c["frame"] = np.arange(n) % 3
    c["key"] = 1000 * rank + np.arange(n) + 7 * step
    c["cls"] = (np.arange(n) + rank) % 3
    c["x"], c["y"], c["w"], c["h"] = rank, step, 5, 9
    c["score_strong"] = rank + np.arange(n) * 0.25
    c["score_weak"] = -1.5 * step
    return c


@pytest.mark.parametrize("world", [1, 2, 3])
def test_native_gather_in_process_group(S, world):
    comms = S.Comm.local_group(world)
    sizes = lambda step: [(5 + step, 0, 3, 11)[(r + step) % 4] for r in range(world)]        # ragged, with empty shares
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            res = []
            for step in range(4):                                   # several rounds over the same communicators
                mine = _fake_cands(S, r, sizes(step)[r], step)
                res.append(comms[r].gather(mine, frame_offset=100 * r))
            out[r] = res
        except Exception as e:                                       # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    assert not errs, errs
    for step in range(4):
        exp = []
        for r in range(world):
            e = _fake_cands(S, r, sizes(step)[r], step)
            e["frame"] += 100 * r
            exp.append(e)
        exp = np.concatenate(exp) if exp else np.zeros(0, S.CAND_DTYPE)
        for r in range(world):
            got, counts = out[r][step]
            assert counts.tolist() == sizes(step)
            assert got.tobytes() == exp.tobytes()          # every rank receives all records, ordered by rank
    for c in comms:
        c.close()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_allgather_bytes_in_process_group(S, world):
    """str_er_comm_allgather_bytes (the strip blobs' transport) over the in-process group: ragged and empty
    contributions, several rounds, every rank receives every rank's bytes."""
    comms = S.Comm.local_group(world)
    blob = lambda r, step: bytes(((7 * r + 3 * step + i) % 251 for i in range((0, 5, 70001, 64)[(r + step) % 4])))
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            out[r] = [comms[r].allgather_bytes(blob(r, step)) for step in range(4)]
        except Exception as e:                                       # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    assert not errs, errs
    for step in range(4):
        for r in range(world):
            assert out[r][step] == [blob(q, step) for q in range(world)]
    for c in comms:
        c.close()


def test_a_failing_rank_fails_every_rank(S):
    """a rank whose own arguments are bad must not leave its peers waiting in the collective -- it announces the failure
    in the exchange of the sizes, and every rank returns an error."""
    import ctypes as C
    world = 3
    comms = S.Comm.local_group(world)
    L = S.load_library()
    rcs = [None] * world

    def rank_main(r):
        out = C.c_void_p()
        starts, sizes = (C.c_int64 * world)(), (C.c_int64 * world)()
        buf = C.create_string_buffer(b"abcdef", 6)
        # rank 1 passes a NULL buffer with a positive size
        rcs[r] = L.str_er_comm_allgather_bytes(comms[r].h, None if r == 1 else buf, 6, 0, 0, C.byref(out), starts, sizes)

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(30)
    assert all(not t.is_alive() for t in th), "a rank is still waiting"
    assert rcs[1] == -1 and all(rc is not None and rc < 0 for rc in rcs), rcs
    # the communicators are still usable afterwards
    res = [None] * world
    th = [threading.Thread(target=lambda r=r: res.__setitem__(r, comms[r].allgather_bytes(bytes([r] * (r + 1))))) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(30)
    assert res[0] == [bytes([q] * (q + 1)) for q in range(world)] and res[0] == res[1] == res[2]
    for c in comms:
        c.close()


def test_native_gather_errors(S):
    import ctypes as C
    L = S.load_library()
    h = C.c_void_p()
    assert L.str_er_comm_local_group(0, C.byref(h)) == -1
    comms = S.Comm.local_group(1)
    with pytest.raises(S.StrErError):
        comms[0].gather_last(type("X", (), {"h": None})())          # no context
    comms[0].close()


_RCCL_SCRIPT = r"""
import sys, tempfile
sys.path.insert(0, %r)
import numpy as np
import str_er_amd as S
uid = S.Comm.unique_id()
assert len(uid) == 128
comm = S.Comm.rccl(0, 0, 1, uid)
sp, wp = S.cascade_io.write_golden(tempfile.mkdtemp())
erf = S.ERFilter(params=S.Params(max_width=640, max_height=480, max_frames=1))
erf.load_cascade(0, sp); erf.load_cascade(1, wp)
frame = S.synth.stext_bgr(S.synth.frame_seed(2), 640, 480)
res = erf.text_detect(frame)
got, counts = comm.gather_last(erf, frame_offset=40)
exp = res.cands.copy()
exp["frame"] += 40
exp["node"] = got["node"]      # (the host copy of the result has node = -1 without WANT_NODES; the device records keep the kept slot)
assert counts.tolist() == [len(exp)] and len(exp) > 0
assert got.tobytes() == exp.tobytes()
got2, _ = comm.gather(res.cands, frame_offset=0)      # host records through the same transport
assert got2.tobytes() == res.cands.tobytes()
comm.close()
print("rccl gather ok", len(exp))
"""


@pytest.mark.gpu
def test_rccl_gather_world_of_one():
    """RCCL transport (librccl.so through dlopen): unique id, communicator, ncclAllGather of counts and records; the records
    come from the device array the detect call left them in (str_er_gather_last) and equal the result's own copy.  In a process
    of its own: RCCL finds its ROCm runtime by name, and a test process that has also imported PyTorch holds two of them."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT % root], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl gather ok" in r.stdout, r.stdout + r.stderr
