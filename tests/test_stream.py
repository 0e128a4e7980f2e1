"""Frame ingest: str_er_stream_* keeps several batches of host frames in flight."""
import numpy as np
import pytest


def test_stream_needs_a_gpu_or_fails_loudly(S):
    """No CPU path here either: without a HIP device the stream cannot be created."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(S.StrErError):
        S.FrameStream(S.Params(max_width=64, max_height=64, max_frames=1), depth=2)


@pytest.mark.gpu
def test_stream_matches_direct_calls(S, cascade_paths):
    W, H, F, D = 320, 240, 2, 3
    prm = S.Params(max_width=W, max_height=H, max_frames=F)
    st = S.FrameStream(prm, depth=D)
    assert st.depth == D
    st.load_cascade(0, cascade_paths[0]); st.load_cascade(1, cascade_paths[1])
    ref = S.ERFilter(params=prm)
    ref.load_cascade(0, cascade_paths[0]); ref.load_cascade(1, cascade_paths[1])
    batches = [np.stack([S.synth.stext_bgr(S.synth.frame_seed(100 + 2 * b + i), W, H) for i in range(F)]) for b in range(7)]
    stages = S.STAGE_ALL | S.STAGE_TRACK | S.STAGE_GROUP
    expected = [ref.text_detect(b, stages) for b in batches]
    got, tickets = [], []
    for i, b in enumerate(batches):
        if st.pending() == D:                              # every buffer in flight: collect the oldest first
            t, r = st.next()
            got.append(r); tickets.append(t)
        if i % 2 == 0:                                     # zero-copy flavour: fill the pinned buffer ourselves
            slot, buf = st.acquire()
            buf[: b.size] = b.reshape(-1)
            st.submit(slot, W, H, F, stages)
        else:
            st.submit_copy(b, stages)
    while st.pending():
        t, r = st.next()
        got.append(r); tickets.append(t)
    assert tickets == list(range(1, len(batches) + 1))     # submission order
    for g, e in zip(got, expected):
        assert g.cands.tobytes() == e.cands.tobytes() and len(g.cands) > 0
        assert g.info.tobytes() == e.info.tobytes()
        assert np.array_equal(g.tracks["tracked"], e.tracks["tracked"])
        assert g.text_ers.tolist() == e.text_ers.tolist()
    # back-pressure and misuse are errors, not hangs
    slots = [st.acquire()[0] for _ in range(D)]
    with pytest.raises(S.StrErError):
        st.acquire()
    with pytest.raises(S.StrErError):
        st.next()                                          # nothing submitted
    with pytest.raises(S.StrErError):
        st.submit(slots[0], 4 * W, 4 * H, F, stages)       # does not fit the staging buffer
    for s in slots:
        st.submit(s, W, H, 1, S.STAGE_ALL)
    for _ in slots:
        st.next()
    # an error inside a batch (frame larger than the context) comes back from next(), the stream stays usable
    slot, buf = st.acquire()
    st.submit(slot, W, H, 1, S.STAGE_ALL | S.STAGE_OCR)    # no SVM model loaded
    with pytest.raises(S.StrErError):
        st.next()
    st.submit_copy(batches[0], S.STAGE_ALL)
    assert st.next()[1].cands.tobytes() == ref.text_detect(batches[0]).cands.tobytes()
    st.close(); ref.close()


def test_batch_slots_setting_round_trips(S):
    """str_er_set_batch_slots (include/str_er.h) returns the previous value; 0 / negative = no limit (CPU: no HIP call involved)."""
    old = S.set_batch_slots(3)
    try:
        assert S.set_batch_slots(-5) == 3
        assert S.set_batch_slots(0) == 0
    finally:
        S.set_batch_slots(old)


@pytest.mark.gpu
def test_batch_slots_do_not_change_results_and_do_not_hang(S, cascade_paths):
    """With one slot the large batches of four contexts take turns on the GPU (a call gives its slot back when its kernels are done, before its host-side work);
    every call returns what the same call returns without the limit, an error inside a call gives the slot back, and calls of a frame or two never wait for one."""
    import threading
    W, H, F = 160, 120, 17                                  # 17 frames x 6 planes = 102 planes: a "large" batch (> 96 planes)
    prm = S.Params(max_width=W, max_height=H, max_frames=F)
    ctxs = []
    for _ in range(4):
        f = S.ERFilter(params=prm)
        f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
        ctxs.append(f)
    batches = [np.stack([S.synth.stext_bgr(S.synth.frame_seed(500 + 31 * b + i), W, H) for i in range(F)]) for b in range(4)]
    expected = [ctxs[0].text_detect(b) for b in batches]
    old = S.set_batch_slots(1)
    try:
        got = [[None] * 3 for _ in range(4)]

        def work(p):
            for k in range(3):
                got[p][k] = ctxs[p].text_detect(batches[(p + k) % 4])
        th = [threading.Thread(target=work, args=(p,)) for p in range(4)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in th), "a call is still waiting for a slot"
        for p in range(4):
            for k in range(3):
                e = expected[(p + k) % 4]
                assert got[p][k].cands.tobytes() == e.cands.tobytes() and len(e.cands) > 0
                assert got[p][k].info.tobytes() == e.info.tobytes()
        with pytest.raises(S.StrErError):                   # (no SVM model loaded: the call fails after it has taken its slot)
            ctxs[1].text_detect(batches[0], S.STAGE_ALL | S.STAGE_OCR)
        assert ctxs[2].text_detect(batches[1]).cands.tobytes() == expected[1].cands.tobytes()      # the slot came back
        assert ctxs[3].text_detect(batches[2][:1]).cands.tobytes() == ctxs[0].text_detect(batches[2][:1]).cands.tobytes()
    finally:
        S.set_batch_slots(old)
        for f in ctxs:
            f.close()
