"""The N > 1 code paths EXECUTED on the one GPU a box has (missing #3): `bench.py --gpus 2` as two processes under
torch.distributed.run sharing device 0 (developer knobs STR_ER_BENCH_FORCE_DEVICE / STR_ER_BENCH_BACKEND=gloo), and BASELINE configs[3]'s
data flow -- 64 frames dealt to 8 ranks, the candidate gather of the C ABI -- with 8 contexts on one device over an in-process group of 8.
No scaling number is claimed from either."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ["frame", "ch", "pyr", "level", "cls", "x", "y", "w", "h", "area", "key", "score_strong", "score_weak"]


def test_bench_two_ranks_on_one_device():
    env = dict(os.environ, STR_ER_BENCH_FORCE_DEVICE="0", STR_ER_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames-per-gpu", "4", "--pipelines", "2",
           "--no-cpu-baseline", "--no-latency", "--no-host-frames", "--no-ties-leg"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    # (a step runs its 4-frame batch as often as it takes for the timed region to last --min-region-s)
    per_step = d["config"]["frames_per_gpu_per_step"]
    assert d["config"]["frames_per_batch"] == 4 and per_step == 4 * d["config"]["batches_per_step"] and d["value"] > 0
    assert "gather" in d["config"]                                     # the exchange ran (torch.distributed over gloo here)
    # whole-job value = frames of BOTH ranks per second of the slowest rank
    assert abs(d["value"] - 2 * per_step * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 0.02


def test_config4_dataflow_eight_ranks_in_one_process(S, cascade_paths):
    """64 frames, 8 per rank, every rank detects its share and gathers over the C ABI (in-process group of 8): every rank ends up with
    the records one 64-frame call gives, frame numbers global.  640x360 frames keep it quick; the flow does not depend on the size."""
    W, H, WORLD, PER = 640, 360, 8, 8
    frames = np.stack([S.synth.stext_bgr(S.synth.frame_seed(i), W, H) for i in range(WORLD * PER)])
    one = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=WORLD * PER))
    one.load_cascade(0, cascade_paths[0]); one.load_cascade(1, cascade_paths[1])
    whole = one.text_detect(frames).cands
    one.close()
    comms = S.Comm.local_group(WORLD)
    out, errs = [None] * WORLD, []

    def rank_main(r):
        try:
            first, n = S.dist.shard_frames(WORLD * PER, r, WORLD)
            f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=PER))
            f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
            mine = f.text_detect(frames[first:first + n]).cands
            out[r] = comms[r].gather(mine, frame_offset=first)
            f.close()
        except Exception as e:                                       # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(WORLD)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not errs, errs
    for r in range(WORLD):
        got, counts = out[r]
        assert int(counts.sum()) == len(whole) and len(counts) == WORLD and (counts > 0).all()
        assert got[FIELDS].tolist() == whole[FIELDS].tolist()
    for c in comms:
        c.close()


def test_config4_at_its_size_eight_contexts_and_the_oracle(S, cascade_paths, oracle, oracle_cascades):
    """BASELINE configs[3] at its stated size (missing #2): a batch of 64 frames of 1920x1080, 8 per rank, through 8 contexts
    (one per rank, all on this box's one device) and the candidate gather of the C ABI (in-process group of 8).  Every rank must end up
    with the records ONE 64-frame call gives, and a sample of the batch -- frames 0, 31 and 63, all six planes of each -- is compared with
    the oracle node table for node table, pool, classes and scores (the reference's loop over the planes: src/ER.cpp:50-60)."""
    from concurrent.futures import ThreadPoolExecutor
    from conftest import check_plane_against_oracle

    W, H, WORLD, PER = 1920, 1080, 8, 8
    N = WORLD * PER
    with ThreadPoolExecutor(8) as ex:
        frames = np.stack(list(ex.map(lambda i: S.synth.stext_bgr(S.synth.frame_seed(i), W, H), range(N))))
    one = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=N))
    one.load_cascade(0, cascade_paths[0]); one.load_cascade(1, cascade_paths[1])
    whole_res = one.text_detect(frames, want_nodes=True)
    whole = whole_res.cands
    assert len(whole_res.planes) == N * 6
    # the oracle on a sample of the batch
    sample = [p for p in whole_res.planes if p.frame in (0, 31, 63)]
    assert len(sample) == 18
    six = {fr: oracle.compute_channels(frames[fr]) for fr in (0, 31, 63)}

    def one_plane(p):
        check_plane_against_oracle(oracle, p, six[p.frame][p.ch], oracle_cascades)
        return p.n_pool
    with ThreadPoolExecutor(8) as ex:
        assert sum(ex.map(one_plane, sample)) > 0
    one.close()
    comms = S.Comm.local_group(WORLD)
    out, errs = [None] * WORLD, []

    def rank_main(r):
        try:
            first, n = S.dist.shard_frames(N, r, WORLD)
            f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=PER))
            f.load_cascade(0, cascade_paths[0]); f.load_cascade(1, cascade_paths[1])
            mine = f.text_detect(frames[first:first + n]).cands
            out[r] = comms[r].gather(mine, frame_offset=first)
            f.close()
        except Exception as e:                                       # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(WORLD)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    assert not errs, errs
    for r in range(WORLD):
        got, counts = out[r]
        assert int(counts.sum()) == len(whole) and len(counts) == WORLD and (counts > 0).all()
        assert got[FIELDS].tolist() == whole[FIELDS].tolist()
    for c in comms:
        c.close()
