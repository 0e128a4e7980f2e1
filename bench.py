#!/usr/bin/env python3
"""bench.py -- frames/sec of the ER hot path (extract + NMS + 2-stage classify) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames-per-gpu F]
                    [--workload pyr3x8|native6] [--kind text|noise] [--no-cpu-baseline]

A "step" is one pass of the hot path over one batch of F synthetic 1920x1080 BGR frames per
GPU that are ALREADY RESIDENT IN HBM: compute_channels (+ pyramid) -> per-plane component
tree -> NMS -> LBP + strong/weak cascades -> candidate records copied back to the host
(and, for N > 1, gathered across ranks with RCCL).  Frames are dealt out to ranks, so per-GPU
work is fixed as N grows ("weak" scaling) and `value` = all frames processed by all ranks
per second.

Workloads (BASELINE.json `configs`):
  pyr3x8  : configs[1]/[2]: planes {Y,Cr,Cb} x 8 pyramid levels (24 planes, 12.40 Mpx/frame)
  native6 : what the reference's text_detect really runs: {Y,Cr,Cb,255-Y,255-Cr,255-Cb} at
            native resolution (6 planes, 12.44 Mpx/frame; src/ER.cpp:114-128)

Extra objects on the JSON line:
  roofline       : HBM roofline of the dominant kernel (k_tile_tree): algorithmic bytes per launch over its ISOLATED
                   launch duration (HIP events recorded by the library on the stream the kernel runs on, one batch in
                   flight, mean of 3 launches after the timed region).  With several batches sharing the GPU an
                   event-to-event time also contains the other batches' kernels (`overlapped_event_ms`).
  cpu_baseline   : the oracle (a plain-C port of the reference's CPU algorithm) timed on this box's host cores on a
                   bounded sample, threads over planes like the reference's `#pragma omp parallel for` (src/ER.cpp:50).
  pcie_inclusive : the same step with the frames starting in page-locked HOST memory (ingest stream, uploads
                   overlapping compute); never the reported `value`.
  latency_1frame : one frame per call, one call in flight: call-to-return wall time.
"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 1920, 1080
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_MEASURED_GBS = 6290.0    # ... and what a float4 copy reaches (same guide)
WORKLOADS = {
    "pyr3x8": dict(n_pyr_levels=8, channel_mask=0x07, label="1920x1080 BGR, {Y,Cr,Cb} x 8 pyramid levels (BASELINE configs[1]/[2])"),
    "native6": dict(n_pyr_levels=1, channel_mask=0x3F, label="1920x1080 BGR, reference-native 6 planes x 1 level"),
}


def effective_cpus() -> int:
    """CPUs this process can keep busy: os.cpu_count() cut down to the container's cgroup CPU quota (the GPU boxes report 256 cores and grant 16)."""
    n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, -(-int(q) // int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = max(1, min(n, -(-q // per)))
        except Exception:
            pass
    return n


def plane_pixels(workload: str) -> int:
    """Sum of plane pixels per frame (12 395 367 for pyr3x8, 12 441 600 for native6)."""
    cfg = WORKLOADS[workload]
    nch = bin(cfg["channel_mask"]).count("1")
    tot = 0
    for k in range(cfg["n_pyr_levels"]):
        s = 2.0 ** (-0.5 * k)
        tot += max(1, int(np.floor(W * s + 0.5))) * max(1, int(np.floor(H * s + 0.5))) * nch
    return tot


def cpu_baseline(kind: str, workload: str, cascades, budget_s: float = 12.0):
    """Oracle on host cores: per frame compute_channels (+pyramid) then one thread per plane."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle import Oracle

    S = importlib.import_module("scene-text-recognition_amd")
    cfg = WORKLOADS[workload]
    o = Oracle()
    cs, cw = o.cascade_load(cascades[0]), o.cascade_load(cascades[1])
    ncores = effective_cpus()

    def planes_of(frame):
        six = o.compute_channels(frame)
        out = []
        for ch in range(6):
            if cfg["channel_mask"] & (1 << ch):
                out.extend(o.pyramid(six[ch], cfg["n_pyr_levels"]))
        return out

    def run_plane(p):
        r = o.detect_plane(p, cs, cw)
        return len(r["pool"])

    frames_done, t_total, pooled = 0, 0.0, 0
    nthreads = 1
    while t_total < budget_s and frames_done < 64:
        frame = S.synth.KINDS[kind](S.synth.frame_seed(frames_done), W, H)
        t0 = time.perf_counter()
        planes = planes_of(frame)
        nthreads = max(1, min(len(planes), ncores))
        with ThreadPoolExecutor(nthreads) as ex:
            pooled += sum(ex.map(run_plane, planes))
        t_total += time.perf_counter() - t0
        frames_done += 1
    return {"value": round(frames_done / t_total, 4), "unit": "frames/s", "cores": nthreads, "kind": "port",
            "sample": f"{frames_done} S-{kind} 1920x1080 frame(s), workload {workload}, {t_total:.1f} s of CPU wall time, "
                      f"oracle/er_oracle.c -O2, one thread per plane ({nthreads} threads; the box reports {os.cpu_count()} cores, its CPU quota grants {ncores})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="timed batches (one step = one batch of --frames-per-gpu frames per GPU; 40 steps = about a third of a second)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-gpu", type=int, default=48,
                    help="frames per batch (= per step) and GPU; 48-64 amortise the per-batch launch latencies best (32 or 96: about 7 %% slower)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="pyr3x8")
    ap.add_argument("--kind", choices=["text", "noise", "ties"], default="text",
                    help="synthetic frames (SURVEY 8d): S-text (default), S-noise (stress), S-ties = S-text with a glyph in every third frame that "
                         "makes an NMS sibling tie whose outcome changes the pool (--ties-every 8: 1.2 %% of the planes need the reference's flood "
                         "order walked, 3: 3.1 %%)")
    ap.add_argument("--ties-every", type=int, default=8, help="S-ties: one tie glyph in every N-th frame")
    ap.add_argument("--no-ties-leg", action="store_true",
                    help="skip the `nms_ties_leg` object of the default run (the same measurement as `value`, on S-ties frames, fewer steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--group", action="store_true",
                    help="also run the rest of text_detect: calc_color + er_track + er_grouping(inner_sup) (SURVEY 8(f) rows 1-2)")
    ap.add_argument("--no-host-frames", action="store_true",
                    help="skip the PCIe-inclusive leg (`pcie_inclusive`: frames start in page-locked host memory and go through the "
                         "ingest stream str_er_stream_*, uploads overlapping compute); it is never the reported `value`")
    ap.add_argument("--sibling-order", type=int, default=0, help="developer knob: 0 = exact NMS ties (default), 2 = largest-key rule (no flood order walk)")
    ap.add_argument("--no-latency", action="store_true", help="skip the 1-frame-per-call latency leg (`latency_1frame`)")
    ap.add_argument("--ocr", action="store_true",
                    help="BASELINE configs[2]: also run the chain-code + SVM character scorer on every strong/weak ER "
                         "(synthetic stand-in for the missing classifier/OCR.model: scene-text-recognition_amd/data/ocr_synth.model.gz)")
    ap.add_argument("--pipelines", type=int, default=6,
                    help="independent batches in flight per GPU (each has its own context, stream and workspace).  Three hide the "
                         "host-side result handling and the low-parallelism tails of a batch; the flood order walk that decides an NMS "
                         "sibling tie (about one plane per 48 S-text frames, 20-50 ms on a host core) needs a few more")
    args = ap.parse_args()

    # (torch first: it brings its own HIP runtime, which the library must share -- loaded the other way round the process has two)
    import torch
    import torch.distributed as dist

    S = importlib.import_module("scene-text-recognition_amd")
    # Several contexts in flight use three HIP streams each (main, alt NMS pass, tie pass); with the runtime's default of 4 hardware queues
    # the long single-workgroup kernels of one context's tie pass sit in front of another context's tile kernel (measured: 4940 -> 5330
    # frames/s with 16).  The library does not edit its host's environment; this application opts in (str_er_apply_runtime_hint) before
    # the process's first HIP call -- the runtime reads the setting when it initialises, which importing torch does not do.
    S.apply_runtime_hint()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path to measure)")
    # developer knobs (single-GPU boxes): run every rank on one device and gather over gloo
    dev_index = int(os.environ.get("STR_ER_BENCH_FORCE_DEVICE", local_rank))
    backend = os.environ.get("STR_ER_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    gather_device = device if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    # the candidate gather: the library's own (RCCL through the C ABI, records taken from the device array of the detect call);
    # if it cannot be set up on every rank, torch.distributed carries the same exchange
    comm = None
    if world > 1:
        ok = 1
        try:
            uid = torch.zeros(128, dtype=torch.uint8, device=gather_device)
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(S.Comm.unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            if backend != "nccl":
                raise RuntimeError("not an RCCL run")
            # (never executed on more than one GPU so far: the communicator is created on a helper thread with a time limit, so that a rank stuck inside
            # ncclCommInitRank -- a collective: then every rank is -- falls back to torch.distributed with the others instead of hanging the run)
            import threading
            box = {}

            def _mk():
                try:
                    box["comm"] = S.Comm.rccl(dev_index, rank, world, uid.cpu().numpy().tobytes())
                except Exception as ee:                           # noqa: BLE001
                    box["err"] = ee
            th = threading.Thread(target=_mk, daemon=True)
            th.start()
            th.join(float(os.environ.get("STR_ER_BENCH_COMM_TIMEOUT", "90")))
            if th.is_alive():
                raise RuntimeError("native RCCL communicator: no answer within the time limit")
            if "err" in box:
                raise box["err"]
            comm = box["comm"]
        except Exception as e:                                    # noqa: BLE001
            print(f"[bench] rank {rank}: native RCCL gather not available ({e}); using torch.distributed", file=sys.stderr, flush=True)
            ok, comm = 0, None
        flag = torch.tensor([ok], dtype=torch.int32, device=gather_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0 and comm is not None:
            comm.close()
            comm = None

    cfg = WORKLOADS[args.workload]
    F = args.frames_per_gpu
    P = max(1, args.pipelines)
    tmp = tempfile.mkdtemp()
    cascades = S.cascade_io.write_golden(tmp)
    filters = []
    for _ in range(P):
        f = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=F, n_pyr_levels=cfg["n_pyr_levels"],
                                       channel_mask=cfg["channel_mask"], device=dev_index, sibling_order=args.sibling_order))
        f.load_cascade(0, cascades[0])
        f.load_cascade(1, cascades[1])
        if args.ocr:
            import gzip
            f.load_svm_model_text(gzip.open(S.cascade_io.ocr_model_path()).read(), 1800)
        filters.append(f)
    # --ocr alone scores every strong/weak ER (slope 0); with --group the scorer runs where er_ocr runs it: on the members of the text lines
    stages = S.STAGE_ALL | ((S.STAGE_TRACK | S.STAGE_GROUP | S.GROUP_INNER_SUP) if args.group else 0)
    if args.ocr:
        stages |= S.STAGE_OCR_LINES if args.group else S.STAGE_OCR

    ws_bytes = filters[0].workspace_bytes()
    # synthetic frames of this rank's shard: F DISTINCT frames, global frame index = rank*F + i, seed = 0x5EED0000 + index
    # (SURVEY 8(d)); generated on host threads (numpy releases the GIL)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(min(F, max(1, (os.cpu_count() or 1) // 2), 16)) as ex:
        make = (lambda sd, w, h: S.synth.sties_bgr(sd, w, h, args.ties_every)) if args.kind == "ties" else S.synth.KINDS[args.kind]
        frames = np.stack(list(ex.map(lambda i: make(S.synth.frame_seed(rank * F + i), W, H), range(F))))
    d_frames = torch.from_numpy(frames).to(device)
    torch.cuda.synchronize()

    # P batches are in flight at once: worker p runs batches p, p+P, p+2P, ... on its own context/stream
    # (the C call releases the GIL); the main thread consumes the results in batch order and does the
    # cross-rank gather, so collectives are issued in the same order on every rank.
    import queue
    import threading

    def tie_totals():
        st = [f.tie_stats() for f in filters]
        return sum(x["planes_walked"] for x in st), sum(x["walk_ms_total"] for x in st), st[0]["host_threads"]

    def run(n_batches, d_in=None):
        d_in = d_frames if d_in is None else d_in
        results = [None] * n_batches
        done = [threading.Event() for _ in range(n_batches)]
        turn = threading.Condition()
        state = {"next": 0}

        def worker(p):
            torch.cuda.set_device(dev_index)
            for i in range(p, n_batches, P):
                results[i] = filters[p].detect_bgr_device(d_in.data_ptr(), W, H, F, stages)
                if comm is not None:
                    # the records are still in this context's device array: gather them before the context takes its next
                    # batch, and in batch order -- every rank issues its collectives in the same order
                    with turn:
                        turn.wait_for(lambda: state["next"] == i)
                    comm.gather_last(filters[p], frame_offset=rank * F)
                    with turn:
                        state["next"] = i + 1
                        turn.notify_all()
                done[i].set()

        threads = [threading.Thread(target=worker, args=(p,)) for p in range(P)]
        for t in threads:
            t.start()
        prof, last = {}, None
        for i in range(n_batches):
            done[i].wait()
            r = results[i]
            results[i] = None
            if world > 1 and comm is None:
                S.dist.gather_candidates(r.cands, gather_device, frame_offset=rank * F)
            for k, v in r.profile.items():
                prof[k] = prof.get(k, 0.0) + v
            last = r
        for t in threads:
            t.join()
        return prof, last

    # W untimed warm-up steps -- on EVERY context: a context's first batch sizes its node records for the frames' content (and on
    # noise-like frames picks the large tile kernel), which must not happen inside the timed region of whichever contexts W did not reach
    run(args.warmup * P)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ties0 = tie_totals()
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    prof_sum, r = run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    host_cpu_per_wall = (time.process_time() - cpu0) / max(elapsed, 1e-9)      # host CPUs this process kept busy during the timed region
    ties1 = tie_totals()
    nms_ties = {"tie_planes_per_batch": round((ties1[0] - ties0[0]) / args.steps, 2),
                "flood_walk_ms_per_batch": round((ties1[1] - ties0[1]) / args.steps, 2), "host_threads": ties1[2],
                "note": "planes of a batch whose NMS sibling tie changes the pool: the reference's flood order is walked on a host core for each "
                        "(host ms summed over planes), on the library's process-wide pool of at most host_threads threads"}
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gather_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # un-overlapped kernel durations: a few single-pipeline steps after the timed region (with P > 1 the
    # events of the timed region include the time a kernel spends sharing the GPU with the other batch)
    serial_prof = {}
    if P > 1:
        n_cal = 3
        for _ in range(n_cal):
            rc = filters[0].detect_bgr_device(d_frames.data_ptr(), W, H, F, stages)
            for k, v in rc.profile.items():
                serial_prof[k] = serial_prof.get(k, 0.0) + v / n_cal
    else:
        serial_prof = {k: v / max(args.steps, 1) for k, v in prof_sum.items()}
    r_tree_stats = filters[0].last_tree_stats()

    # ties leg: the same measurement on tie-rich frames (S-ties), so that the cost of exactness is on the line
    ties_leg = None
    if not args.no_ties_leg and args.kind == "text" and world == 1 and args.sibling_order == 0 and not args.ocr and not args.group:
        with ThreadPoolExecutor(min(F, max(1, (os.cpu_count() or 1) // 2), 16)) as ex:
            tf = np.stack(list(ex.map(lambda i: S.synth.sties_bgr(S.synth.frame_seed(i), W, H, args.ties_every), range(F))))
        d_ties = torch.from_numpy(tf).to(device)
        torch.cuda.synchronize()
        run(P, d_ties)
        n_t = max(P, args.steps // 2)
        a0 = tie_totals()
        cpu1 = time.process_time()
        t1 = time.perf_counter()
        run(n_t, d_ties)
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        ties_cpu = (time.process_time() - cpu1) / max(el, 1e-9)
        a1 = tie_totals()
        ties_leg = {"value": round(F * n_t / el, 2), "unit": "frames/s", "steps": n_t, "ms_per_step": round(1e3 * el / n_t, 3),
                    "tie_planes_per_batch": round((a1[0] - a0[0]) / n_t, 2), "tie_plane_share": round((a1[0] - a0[0]) / n_t / (F * bin(cfg['channel_mask']).count('1') * cfg['n_pyr_levels']), 4),
                    "flood_walk_ms_per_batch": round((a1[1] - a0[1]) / n_t, 2), "host_threads": a1[2], "host_cores": os.cpu_count(), "host_cpu_quota": effective_cpus(), "host_cpus_busy": round(ties_cpu, 2),
                    "note": f"S-ties frames (S-text + one double-L glyph in every {args.ties_every}th frame: an NMS sibling tie with two different outcomes); same "
                            "batches in flight as `value`; flood_walk_ms = host time of the reference-order walks, summed over planes.  The walks are "
                            "bound by host memory latency (~6 ms per 1920x1080 plane): once tie planes per batch x 6 ms / host_threads exceeds the "
                            "GPU's time per batch the leg is host-bound (measured with a glyph in every 3rd frame, 3.1 % of the planes: 0.47 of `value`)"}
        del d_ties

    # latency leg: ONE frame per call, one batch in flight (north_star: ">= 500 fps end-to-end on 1920x1080" is a
    # per-frame statement; the headline `value` needs 48-frame batches x 3 in flight)
    latency = None
    if not args.no_latency and rank == 0:
        for f in filters[1:]:
            f.close()
        filters = filters[:1]
        f1 = S.ERFilter(params=S.Params(max_width=W, max_height=H, max_frames=1, n_pyr_levels=cfg["n_pyr_levels"],
                                        channel_mask=cfg["channel_mask"], device=dev_index, sibling_order=args.sibling_order))
        f1.load_cascade(0, cascades[0]); f1.load_cascade(1, cascades[1])
        if args.ocr:
            import gzip
            f1.load_svm_model_text(gzip.open(S.cascade_io.ocr_model_path()).read(), 1800)
        fb = frames[0].nbytes
        for i in range(3):
            f1.detect_bgr_device(d_frames.data_ptr() + (i % F) * fb, W, H, 1, stages)
        n_lat = 40
        dev_ms, host_ms = [], []
        for i in range(n_lat):
            t1 = time.perf_counter()
            f1.detect_bgr_device(d_frames.data_ptr() + (i % F) * fb, W, H, 1, stages)
            dev_ms.append(1e3 * (time.perf_counter() - t1))
        for i in range(n_lat):
            t1 = time.perf_counter()
            f1.text_detect(frames[i % F], stages)          # pageable host frame: H2D copy inside the call
            host_ms.append(1e3 * (time.perf_counter() - t1))
        f1.close()
        latency = {"ms_per_frame": round(float(np.median(dev_ms)), 3), "frames_per_s": round(1e3 / float(np.median(dev_ms)), 1),
                   "ms_per_frame_p90": round(float(np.percentile(dev_ms, 90)), 3),
                   "ms_per_frame_host_input": round(float(np.median(host_ms)), 3),
                   "frames_per_s_host_input": round(1e3 / float(np.median(host_ms)), 1),
                   "note": f"1 frame per call, 1 call in flight, call-to-return wall time incl. the candidate copy to the host; median of {n_lat}; "
                           "host_input = the frame starts in pageable host memory (H2D inside the call)"}

    pcie = pcie_nv12 = None
    if not args.no_host_frames and not args.ocr and rank == 0:
        # "a frame = BGR upload excluded and included (both reported)".  The frames sit in the stream's page-locked
        # staging buffers (where a decoder would put them); every step uploads its 3*W*H*F bytes again.
        for f in filters:
            f.close()
        st = S.FrameStream(S.Params(max_width=W, max_height=H, max_frames=F, n_pyr_levels=cfg["n_pyr_levels"],
                                    channel_mask=cfg["channel_mask"], device=dev_index, sibling_order=args.sibling_order), depth=P)
        st.load_cascade(0, cascades[0]); st.load_cascade(1, cascades[1])

        def stream_steps(n):
            for _ in range(n):
                if st.pending() == P:
                    st.next()
                slot, buf = st.acquire()
                if not filled[slot]:
                    buf[: frames.size] = frames.reshape(-1)
                    filled[slot] = True
                st.submit(slot, W, H, F, stages)
            while st.pending():
                st.next()

        # what the host link of this box delivers: the same bytes, page-locked memory -> device, nothing else running
        pin = torch.empty(frames.size, dtype=torch.uint8).pin_memory()
        dst = torch.empty(frames.size, dtype=torch.uint8, device=device)
        dst.copy_(pin, non_blocking=True); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            dst.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()
        h2d_gbs = 5 * frames.size / (time.perf_counter() - t1) / 1e9
        del pin, dst
        filled = [False] * P
        stream_steps(max(P, args.warmup))
        t1 = time.perf_counter()
        stream_steps(args.steps)
        el = time.perf_counter() - t1
        pcie = {"value": round(F * args.steps / el, 2), "unit": "frames/s (this rank)", "ms_per_step": round(1e3 * el / args.steps, 3),
                "h2d_bytes_per_step": int(frames.size), "h2d_gbs": round(frames.size * args.steps / el / 1e9, 2),
                # (a torch copy of the same bytes from a torch-pinned tensor, nothing else running: 57 GB/s on most of the round's boxes, 26 on one where the
                # stream itself -- hipHostMalloc'ed staging, several copies queued -- moved 43: the larger of the two is what the link is known to deliver)
                "h2d_gbs_link_alone": round(h2d_gbs, 2), "frames_per_s_at_link_rate": round(max(h2d_gbs, frames.size * args.steps / el / 1e9) * 1e9 / (frames.size / F), 1),
                "note": "host BGR frames in page-locked memory -> str_er_stream (upload of one batch overlaps the kernels of the others)"}
        # ... and the same frames as a video decoder would deliver them: NV12, half the bytes (build-defined ingest, include/str_er.h)
        with ThreadPoolExecutor(min(F, max(1, (os.cpu_count() or 1) // 2), 16)) as ex:
            nv = np.stack(list(ex.map(S.synth.nv12_from_bgr, frames)))

        def stream_steps_nv12(n):
            for _ in range(n):
                if st.pending() == P:
                    st.next()
                slot, buf = st.acquire()
                if not filled[slot]:
                    buf[: nv.size] = nv.reshape(-1)
                    filled[slot] = True
                st.submit_nv12(slot, W, H, F, stages)
            while st.pending():
                st.next()

        filled = [False] * P
        stream_steps_nv12(max(P, args.warmup))
        t1 = time.perf_counter()
        stream_steps_nv12(args.steps)
        el = time.perf_counter() - t1
        pcie_nv12 = {"value": round(F * args.steps / el, 2), "unit": "frames/s (this rank)", "ms_per_step": round(1e3 * el / args.steps, 3),
                     "h2d_bytes_per_step": int(nv.size), "h2d_gbs": round(nv.size * args.steps / el / 1e9, 2),
                     "note": "the same frames as NV12 (luma + interleaved Cb/Cr at half resolution: what a decoder delivers) through the same stream; "
                             "the NV12 -> Y/Cr/Cb step is build-defined (chroma replicated 2x2), so the planes -- and the candidates -- are not "
                             "those of the BGR frames"}
        st.close()

    if rank == 0:
        total_frames = F * world * args.steps
        fps = total_frames / elapsed
        px = plane_pixels(args.workload)
        n_pool = len(r.cands)
        # algorithmic bytes per frame, SURVEY.md 8(d): 3WH (BGR read) + 2*sum(plane px) (each 8-bit
        # plane written once, read once) + N_pool*(2704+48)
        b_alg = 3 * W * H + 2 * px + (n_pool / F) * (2704 + 48)
        # dominant kernel: k_tile_tree reads every plane pixel once -> px bytes per frame.  Its duration is the ISOLATED one
        # (one batch in flight, HIP events on the library's stream around the launch): with P batches sharing the GPU an
        # event-to-event time also contains the other batches' kernels and is not a per-launch cost.
        tile_ms_overlapped = prof_sum.get("tile_tree", 0.0) / max(args.steps, 1)
        tile_ms = serial_prof.get("tile_tree", 0.0) or tile_ms_overlapped
        tile_bytes = px * F
        achieved = tile_bytes / (tile_ms * 1e-3) / 1e9 if tile_ms > 0 else 0.0
        traffic, traffic_source = None, "not measured in this run (no rocprofv3 counter pass)"
        pmc = os.path.join(ROOT, "profiles", "pmc_tile_tree.json")
        if os.path.exists(pmc):
            try:
                with open(pmc) as fh:
                    j = json.load(fh)
                if j.get("workload") == args.workload and j.get("frames_per_launch"):
                    traffic = j["hbm_bytes_per_launch"] * F / j["frames_per_launch"]
                    traffic_source = ("STORED figure, not measured in this run: profiles/pmc_tile_tree.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
                                      f"passes of this command, collected {j.get('collected', 'earlier')}; FETCH doubled per the gfx950 note of MI355X_MICROARCH.md), scaled to "
                                      f"{F} frames per launch")
            except Exception:
                traffic = None
        # the other passes of the component tree, priced the same way: algorithmic bytes = what the pass has to read and write once
        # (32-byte node records + their 4-byte counters, two 16-bit seam entries per pixel pair across a tile border) over the pass's isolated time
        ts = r_tree_stats
        tree_bytes = {"group": 2 * 36 * ts["records"], "seam": 4 * ts["seam_pairs"], "resolve": 2 * 36 * ts["records"], "accumulate": 2 * 36 * ts["records"]}
        tree_roof = {}
        for k, nb in tree_bytes.items():
            ms_k = serial_prof.get(k, 0.0)
            if ms_k > 0:
                gbs = nb / (ms_k * 1e-3) / 1e9
                tree_roof[k] = {"ms": round(ms_k, 4), "bytes": int(nb), "achieved": round(gbs, 2), "frac": round(gbs / HBM_PEAK_GBS, 5)}
        tree_roof["note"] = (f"k_group_merge / k_seam / k_resolve / k_reduce of one batch ({ts['records']} exported node records, {ts['seam_pairs']} border pixel pairs, "
                             f"{ts['tiles']} tiles): isolated times as for k_tile_tree; bound by LDS / device-scope atomic latency, not by HBM (DESIGN 3.2)")
        out = {
            "metric": "frames/sec (ER extract + 2-stage classify), 1920x1080",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {cfg['label']}; S-{args.kind} frames" +
                                   ("; + chain-code/SVM OCR scorer on strong+weak ERs (configs[2])" if args.ocr and not args.group else "") +
                                   ("; + calc_color, er_track, er_grouping (text lines)" if args.group else "") +
                                   ("; + chain-code/SVM OCR scorer on the line members with the line slope (er_ocr, configs[2])" if args.ocr and args.group else ""),
                       "frames_per_gpu_per_step": F,
                       "planes_per_frame": bin(cfg['channel_mask']).count('1') * cfg['n_pyr_levels'],
                       "plane_pixels_per_frame": px, "thresh_step": 8, "min_area": 120, "parallelism": f"frames sharded over {world} GPU(s)",
                       "pooled_per_frame": round(n_pool / F, 1), "batches_in_flight": P, "host_cpus_busy": round(host_cpu_per_wall, 2),
                       "workspace_bytes_per_batch_in_flight": ws_bytes, "nms_sibling_ties": "exact (reference flood order)" if args.sibling_order == 0 else "key rule",
                       **({"gather": "RCCL through the C ABI (str_er_gather_last)" if comm is not None else "torch.distributed all_gather"} if world > 1 else {})},
            "nms_ties": nms_ties,
            **({"nms_ties_leg": ties_leg} if ties_leg else {}),
            **({"pcie_inclusive": pcie} if pcie else {}),
            **({"pcie_inclusive_nv12": pcie_nv12} if pcie_nv12 else {}),
            **({"latency_1frame": latency} if latency else {}),
            "roofline": {"bound": "hbm", "kernel": "k_tile_tree", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "frac_of_measured_peak": round(achieved / HBM_MEASURED_GBS, 5),
                         "measured_peak": HBM_MEASURED_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "bytes_per_launch": tile_bytes, "avg_launch_ms": round(tile_ms, 4),
                         "timing": "isolated launch: HIP events on the library's stream, one batch in flight, mean of 3 launches after the timed region",
                         "overlapped_event_ms": round(tile_ms_overlapped, 4),
                         "path_bytes_per_frame": int(b_alg), "path_frac": round(b_alg * fps / world / (HBM_PEAK_GBS * 1e9), 5)},
            "tree_passes_roofline": tree_roof,
            "gpu_ms_per_step_by_kernel_group": {k: round(v / args.steps, 4) for k, v in prof_sum.items()},
            "gpu_ms_per_step_by_kernel_group_serial": {k: round(v, 4) for k, v in serial_prof.items()},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.kind, args.workload, cascades)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
