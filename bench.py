#!/usr/bin/env python3
"""bench.py -- frames/sec of the ER hot path (extract + NMS + 2-stage classify) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames-per-gpu F] [--size 1080p|4k]
                    [--workload pyr3x8|native6|pyr3x12] [--kind text|noise|ties] [--no-cpu-baseline] ...

A "step" is one pass of the hot path over one batch of F synthetic BGR frames per GPU that are ALREADY
RESIDENT IN HBM: compute_channels (+ pyramid) -> per-plane component tree -> NMS -> LBP + strong/weak
cascades -> candidate records copied back to the host (and, for N > 1, gathered across ranks with RCCL).
Frames are dealt out to ranks, so per-GPU work is fixed as N grows ("weak" scaling) and `value` = all
frames processed by all ranks per second.  The timed region of K steps is run `--repeats` times (default 3),
each bracketed by barrier + synchronize; `value` / `ms_per_step` are the median region, `value_min/max` the others.

Workloads (BASELINE.json `configs`):
  pyr3x8  : configs[1] -- the metric's: 1920x1080, planes {Y,Cr,Cb} x 8 pyramid levels (24 planes, 12.40 Mpx/frame)
  native6 : what the reference's text_detect really runs: {Y,Cr,Cb,255-Y,255-Cr,255-Cb} at
            native resolution (6 planes, 12.44 Mpx/frame; src/ER.cpp:114-128)
  pyr3x12 : configs[4]: 3840x2160, {Y,Cr,Cb} x 12 pyramid levels (36 planes, 49.75 Mpx/frame); --size 4k

Extra objects on the JSON line (single-GPU default run):
  roofline         : HBM roofline of the dominant kernel (k_tile_tree): algorithmic bytes per launch over its ISOLATED
                     launch duration (HIP events recorded by the library on the stream the kernel runs on, one batch in
                     flight, mean of 3 launches after the timed region).
  cpu_baseline     : the oracle (a plain-C port of the reference's CPU algorithm) timed on this box's host cores on a
                     bounded sample, threads over planes like the reference's `#pragma omp parallel for` (src/ER.cpp:50).
  config3_ocr_leg  : BASELINE configs[2]: the same batches with the chain-code + SVM character scorer on every strong /
                     weak ER (src/OCR.cpp:67-140): frames/s, ERs scored, per-kernel GPU ms, the MFMA roofline of the RBF
                     kernel-matrix GEMM and its own CPU baseline.  `group_ocr_leg`: the reference's real call pattern
                     (er_track + er_grouping, then the scorer on the members of the text lines, src/ER.cpp:695-747).
  config5_4k_leg   : BASELINE configs[4] on this one GPU: 3840x2160 x 12 levels, frames/s, tile-kernel roofline, latency.
  nms_ties_leg     : the same measurement on frames with NMS sibling ties (cost of exactness).
  pcie_inclusive   : the same step with the frames starting in page-locked HOST memory; never the reported `value`.
  latency_1frame   : one frame per call, one call in flight: call-to-return wall time.
"""
import argparse
import gzip
import importlib
import json
import os
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 1920, 1080
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_MEASURED_GBS = 6290.0    # ... and what a float4 copy reaches (same guide)
MFMA_F32_PEAK_TFLOPS = 157.3  # same guide: dense f32 matrix peak (v_mfma_f32_32x32x2_f32: 256 flop/cycle/CU x 256 CUs x 2.4 GHz)
MFMA_I8_PEAK_TOPS = 5000.0     # same guide: 8-bit matrix instructions run at twice the bf16 rate (microbenchmark ceiling there: 3944 TOPS)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 matrix peak (v_mfma_f32_32x32x16_bf16)
WORKLOADS = {
    "pyr3x8": dict(n_pyr_levels=8, channel_mask=0x07, label="1920x1080 BGR, {Y,Cr,Cb} x 8 pyramid levels (BASELINE configs[1])"),
    "native6": dict(n_pyr_levels=1, channel_mask=0x3F, label="1920x1080 BGR, reference-native 6 planes x 1 level"),
    "pyr3x12": dict(n_pyr_levels=12, channel_mask=0x07, label="3840x2160 BGR, {Y,Cr,Cb} x 12 pyramid levels (BASELINE configs[4])"),
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


def plane_pixels(workload: str, w: int = None, h: int = None) -> int:
    """Sum of plane pixels per frame (12 395 367 for pyr3x8, 12 441 600 for native6, 49 752 702 for pyr3x12)."""
    w, h = (W if w is None else w), (H if h is None else h)
    cfg = WORKLOADS[workload]
    nch = bin(cfg["channel_mask"]).count("1")
    tot = 0
    for k in range(cfg["n_pyr_levels"]):
        s = 2.0 ** (-0.5 * k)
        tot += max(1, int(np.floor(w * s + 0.5))) * max(1, int(np.floor(h * s + 0.5))) * nch
    return tot


def cpu_baseline(kind: str, workload: str, cascades, budget_s: float = 12.0, ocr_model: str = None):
    """Oracle on host cores: per frame compute_channels (+pyramid) then one thread per plane.  With ocr_model: the chain-code
    features + libsvm probability of every strong / weak ER as well (config 3)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle import Oracle, OracleSVM

    S = importlib.import_module("scene-text-recognition_amd")
    cfg = WORKLOADS[workload]
    o = Oracle()
    cs, cw = o.cascade_load(cascades[0]), o.cascade_load(cascades[1])
    svm = OracleSVM(o, ocr_model) if ocr_model else None
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
        scored = 0
        if svm is not None:
            t = r["tree"].nodes
            for j, i in enumerate(r["pool"]):
                if int(r["cls"][j]) == 0:
                    continue
                x, y, w, h = int(t[i]["x"]), int(t[i]["y"]), int(t[i]["w"]), int(t[i]["h"])
                q = o.chain_features(np.ascontiguousarray(p[y:y + h, x:x + w]))
                svm.predict_probability(q / 255.0)
                scored += 1
        return len(r["pool"]), scored

    frames_done, t_total, pooled, scored = 0, 0.0, 0, 0
    nthreads = 1
    while t_total < budget_s and frames_done < 64:
        frame = S.synth.KINDS[kind](S.synth.frame_seed(frames_done), W, H)
        t0 = time.perf_counter()
        planes = planes_of(frame)
        nthreads = max(1, min(len(planes), ncores))
        with ThreadPoolExecutor(nthreads) as ex:
            for a, b in ex.map(run_plane, planes):
                pooled += a
                scored += b
        t_total += time.perf_counter() - t0
        frames_done += 1
    out = {"value": round(frames_done / t_total, 4), "unit": "frames/s", "cores": nthreads, "kind": "port",
           "sample": f"{frames_done} S-{kind} {W}x{H} frame(s), workload {workload}, {t_total:.1f} s of CPU wall time, "
                     f"oracle/er_oracle.c -O2, one thread per plane ({nthreads} threads; the box reports {os.cpu_count()} cores, its CPU quota grants {ncores})"}
    if svm is not None:
        out["sample"] += f"; + oracle chain-code features and oracle/svm_oracle.c on the {scored} strong/weak ERs of those frames, inside the plane's thread"
        out["ers_scored"] = scored
    return out


class Rig:
    """P contexts of one geometry (each with its own stream and workspace), fed by P host threads."""

    def __init__(self, S, P, w, h, F, cfg, dev_index, sibling_order, cascades, svm_text=None):
        self.S, self.P, self.w, self.h, self.F, self.dev_index = S, P, w, h, F, dev_index
        self.filters = []
        for _ in range(P):
            f = S.ERFilter(params=S.Params(max_width=w, max_height=h, max_frames=F, n_pyr_levels=cfg["n_pyr_levels"],
                                           channel_mask=cfg["channel_mask"], device=dev_index, sibling_order=sibling_order))
            f.load_cascade(0, cascades[0])
            f.load_cascade(1, cascades[1])
            if svm_text is not None:
                f.load_svm_model_text(svm_text, 1800)
            self.filters.append(f)

    def close(self):
        for f in self.filters:
            f.close()
        self.filters = []

    def tie_totals(self):
        st = [f.tie_stats() for f in self.filters]
        return sum(x["planes_walked"] for x in st), sum(x["walk_ms_total"] for x in st), st[0]["host_threads"]

    def run(self, n_batches, d_in, stages, comm=None, rank=0, world=1, gather_device=None):
        """P batches are in flight at once: worker p runs batches p, p+P, p+2P, ... on its own context/stream (the C call
        releases the GIL); the main thread consumes the results in batch order and does the cross-rank gather, so collectives
        are issued in the same order on every rank."""
        import torch
        S, P, F = self.S, self.P, self.F
        results = [None] * n_batches
        done = [threading.Event() for _ in range(n_batches)]
        turn = threading.Condition()
        state = {"next": 0}

        def worker(p):
            torch.cuda.set_device(self.dev_index)
            for i in range(p, n_batches, P):
                results[i] = self.filters[p].detect_bgr_device(d_in.data_ptr(), self.w, self.h, F, stages)
                if comm is not None:
                    # the records are still in this context's device array: gather them before the context takes its next
                    # batch, and in batch order -- every rank issues its collectives in the same order
                    with turn:
                        turn.wait_for(lambda: state["next"] == i)
                    comm.gather_last(self.filters[p], frame_offset=rank * F)
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

    def timed(self, n_batches, d_in, stages):
        import torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        prof, last = self.run(n_batches, d_in, stages)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, prof, last

    def serial_profile(self, d_in, stages, n_cal=3):
        """un-overlapped kernel durations: single-context steps (with P > 1 the events of a timed region include the time a
        kernel spends sharing the GPU with the other batches)"""
        sp, last = {}, None
        self.filters[0].set_profiling(True)
        for _ in range(n_cal):
            last = self.filters[0].detect_bgr_device(d_in.data_ptr(), self.w, self.h, self.F, stages)
            for k, v in last.profile.items():
                sp[k] = sp.get(k, 0.0) + v / n_cal
        self.filters[0].set_profiling(False)
        return sp, last

    def overlapped_profile(self, d_in, stages, n_batches):
        """the same events with every context busy (a short region of its own: the timed regions run without the per-group events, which cost stream time)"""
        for f in self.filters:
            f.set_profiling(True)
        prof, _ = self.run(n_batches, d_in, stages)
        for f in self.filters:
            f.set_profiling(False)
        return {k: v / n_batches for k, v in prof.items()}


def make_frames(S, kind, w, h, F, first=0, ties_every=8):
    from concurrent.futures import ThreadPoolExecutor
    make = (lambda sd, ww, hh: S.synth.sties_bgr(sd, ww, hh, ties_every)) if kind == "ties" else S.synth.KINDS[kind]
    with ThreadPoolExecutor(min(F, max(1, (os.cpu_count() or 1) // 2), 16)) as ex:
        return np.stack(list(ex.map(lambda i: make(S.synth.frame_seed(first + i), w, h), range(F))))


def tile_roofline(px, F, tile_ms, tile_ms_overlapped, workload, t2_ms=0.0, t2_share=0.0):
    """The tile trees of a batch are built by two kernels since round 6 -- k_tile_tree2 (level by level on bit masks: the chroma planes) and
    k_tile_tree (pieces + union-find in LDS: the luma planes, and the tiles the first one hands back) -- which between them read every plane
    pixel exactly once: tile_ms is the time of both launches (t2_ms of it k_tile_tree2's, which read t2_share of the pixels)."""
    tile_bytes = px * F
    achieved = tile_bytes / (tile_ms * 1e-3) / 1e9 if tile_ms > 0 else 0.0
    traffic, traffic_source = None, "not measured in this run (no rocprofv3 counter pass)"
    pmc = os.path.join(ROOT, "profiles", "pmc_tile_tree.json")
    if os.path.exists(pmc):
        try:
            with open(pmc) as fh:
                j = json.load(fh)
            if j.get("workload") == workload and j.get("frames_per_launch"):
                traffic = j["hbm_bytes_per_launch"] * F / j["frames_per_launch"]
                traffic_source = ("STORED figure, not measured in this run: profiles/pmc_tile_tree.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
                                  f"passes of this command, collected {j.get('collected', 'earlier')}; FETCH doubled per the gfx950 note of MI355X_MICROARCH.md), scaled to "
                                  f"{F} frames per launch")
        except Exception:
            traffic = None
    per_kernel = None
    if t2_ms > 0 and tile_ms > t2_ms:
        b2, b1, t1_ms = tile_bytes * t2_share, tile_bytes * (1.0 - t2_share), tile_ms - t2_ms
        per_kernel = {"k_tile_tree2": {"avg_launch_ms": round(t2_ms, 4), "bytes_per_launch": int(b2), "achieved": round(b2 / (t2_ms * 1e-3) / 1e9, 2),
                                       "frac": round(b2 / (t2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "planes": "chroma (ch % 3 != 0)"},
                      "k_tile_tree": {"avg_launch_ms": round(t1_ms, 4), "bytes_per_launch": int(b1), "achieved": round(b1 / (t1_ms * 1e-3) / 1e9, 2),
                                      "frac": round(b1 / (t1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "planes": "luma + the tiles k_tile_tree2 hands back (k_tile_tree_fb)"}}
    return {"bound": "hbm", "kernel": "k_tile_tree2 + k_tile_tree (the tile trees: every plane pixel read once, by one of the two)" if per_kernel else "k_tile_tree",
            **({"per_kernel": per_kernel} if per_kernel else {}),
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "frac_of_measured_peak": round(achieved / HBM_MEASURED_GBS, 5),
            "measured_peak": HBM_MEASURED_GBS, "traffic": traffic, "traffic_source": traffic_source,
            "bytes_per_launch": tile_bytes, "avg_launch_ms": round(tile_ms, 4),
            "timing": "isolated launch: HIP events on the library's stream, one batch in flight, mean of 3 launches after the timed region",
            "overlapped_event_ms": round(tile_ms_overlapped, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="timed batches per region (one step = one batch of --frames-per-gpu frames per GPU; 40 steps = about a fifth of a second)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--min-region-s", type=float, default=0.5,
                    help="a timed region lasts at least this long whatever --steps says: a step then runs its batch several times over (config.batches_per_step; "
                         "frames_per_gpu_per_step counts all of them).  A region of 20 single-batch steps is 0.1 s, which one box-to-box hiccup moves by 10 %%")
    ap.add_argument("--repeats", type=int, default=3, help="the timed region of --steps steps is run this many times; `value` is the median region")
    ap.add_argument("--frames-per-gpu", type=int, default=None,
                    help="frames per batch (= per step) and GPU; default 32 for pyr3x8 (round 6, six batches in flight: 16 / 24 / 32 / 40 / 48 frames -> 12.0 / 12.5 / 12.6 / 10.4 / 11.1-11.3 k frames/s; "
                         "until round 5 48-64 were best), 48 for native6 (15.8 k against 15.3 k with 32), 12 for --size 4k (8: -1 %%)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None)
    ap.add_argument("--size", choices=["1080p", "4k"], default="1080p",
                    help="frame size: 1080p = 1920x1080 (the metric's), 4k = 3840x2160 with the 12-level pyramid (BASELINE configs[4]; 12 frames per batch by default)")
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
                    help="BASELINE configs[2] as the MAIN timed region: also run the chain-code + SVM character scorer on every strong/weak ER "
                         "(synthetic stand-in for the missing classifier/OCR.model: scene-text-recognition_amd/data/ocr_synth120.model.gz)")
    ap.add_argument("--no-ocr-legs", action="store_true", help="skip `config3_ocr_leg` / `group_ocr_leg` of the default run")
    ap.add_argument("--no-4k-leg", action="store_true", help="skip `config5_4k_leg` of the default run")
    ap.add_argument("--batch-slots", type=int, default=5,
                    help="str_er_set_batch_slots: at most this many batches have their kernels on the GPU at a time (0: no limit).  Equal batches that share the GPU "
                         "evenly finish together and then wait together for their host side (the flood order walk of an NMS tie) with the GPU idle: with 48-frame "
                         "batches 11.1 k frames/s without a limit, 12.3 k with 3; with the default 32-frame batches it makes no difference on the round's boxes")
    ap.add_argument("--pipelines", type=int, default=7,
                    help="independent batches in flight per GPU (each has its own context, two streams and workspace; default 7 with 5 batch slots since round 6 -- 13.9 k frames/s "
                         "against 13.55-13.6 k for 6 with 4 slots).  Three hide the "
                         "host-side result handling and the low-parallelism tails of a batch; the flood order walk that decides an NMS "
                         "sibling tie (about one plane per 48 S-text frames, about 5 ms on a host core at 1080p, 27 ms at 4K) needs a few more")
    args = ap.parse_args()
    global W, H
    if args.size == "4k":
        W, H = 3840, 2160
    if args.workload is None:
        args.workload = "pyr3x12" if args.size == "4k" else "pyr3x8"
    if args.frames_per_gpu is None:
        args.frames_per_gpu = 12 if args.size == "4k" else (32 if args.workload == "pyr3x8" else 48)

    # (torch first: it brings its own HIP runtime, which the library must share -- loaded the other way round the process has two)
    import torch
    import torch.distributed as dist

    S = importlib.import_module("scene-text-recognition_amd")
    # Several contexts in flight use two HIP streams each (main, alt NMS pass; until round 6 a third for the tie pass); with the runtime's default of 4 hardware queues
    # the long single-workgroup kernels of one context's tie pass sit in front of another context's tile kernel (measured: 4940 -> 5330
    # frames/s with 16).  The library does not edit its host's environment; this application opts in (str_er_apply_runtime_hint) before
    # the process's first HIP call -- the runtime reads the setting when it initialises, which importing torch does not do.
    S.apply_runtime_hint()
    if args.batch_slots >= 0:
        S.set_batch_slots(args.batch_slots)
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
    # config 3's SVM: the stand-in at the reference's training-set size (120 samples per class, src/utils.cpp:1478-1541: 4299 support vectors); the small one of
    # rounds 1-5 (5 per class, 319 support vectors) is timed beside it
    svm_text_small = gzip.open(S.cascade_io.ocr_model_path(5)).read()
    svm_text = gzip.open(S.cascade_io.ocr_model_path(120)).read()
    svm_path = os.path.join(tmp, "ocr_synth.model")
    with open(svm_path, "wb") as fh:
        fh.write(svm_text)
    single = world == 1 and rank == 0
    ocr_legs = single and not args.no_ocr_legs and not args.ocr and not args.group and args.kind == "text" and args.size == "1080p" and args.workload == "pyr3x8"
    rig = Rig(S, P, W, H, F, cfg, dev_index, args.sibling_order, cascades, svm_text if (args.ocr or ocr_legs) else None)
    # --ocr alone scores every strong/weak ER (slope 0); with --group the scorer runs where er_ocr runs it: on the members of the text lines
    st_group = S.STAGE_TRACK | S.STAGE_GROUP | S.GROUP_INNER_SUP
    stages = S.STAGE_ALL | (st_group if args.group else 0)
    if args.ocr:
        stages |= S.STAGE_OCR_LINES if args.group else S.STAGE_OCR

    ws_bytes = rig.filters[0].workspace_bytes()
    # synthetic frames of this rank's shard: F DISTINCT frames, global frame index = rank*F + i, seed = 0x5EED0000 + index
    # (SURVEY 8(d)); generated on host threads (numpy releases the GIL)
    frames = make_frames(S, args.kind, W, H, F, first=rank * F, ties_every=args.ties_every)
    d_frames = torch.from_numpy(frames).to(device)
    torch.cuda.synchronize()

    # W untimed warm-up steps -- on EVERY context: a context's first batch sizes its node records for the frames' content (and on
    # noise-like frames picks the large tile kernel), which must not happen inside the timed region of whichever contexts W did not reach
    rig.run(args.warmup * P, d_frames, stages, comm, rank, world, gather_device)
    # batches per step: enough of them that the K steps of a region last --min-region-s (measured here on 2 P batches; the same on every rank)
    inner = 1
    if args.min_region_s > 0:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rig.run(2 * P, d_frames, stages, comm, rank, world, gather_device)
        torch.cuda.synchronize()
        t_batch = (time.perf_counter() - t0) / (2 * P)
        inner = max(1, int(np.ceil(args.min_region_s / max(args.steps * t_batch, 1e-9))))
        if world > 1:
            t = torch.tensor([inner], dtype=torch.int64, device=gather_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            inner = int(t.item())
    NB = args.steps * inner          # batches per timed region
    regions = []
    prof_sum, r = {}, None
    ties0 = rig.tie_totals()
    host_cpu = []
    for _ in range(max(1, args.repeats)):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        cpu0 = time.process_time()
        t0 = time.perf_counter()
        prof_sum, r = rig.run(NB, d_frames, stages, comm, rank, world, gather_device)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        host_cpu.append((time.process_time() - cpu0) / max(el, 1e-9))     # host CPUs this process kept busy during the timed region
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=gather_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        regions.append(el)
    ties1 = rig.tie_totals()
    n_reg = len(regions)
    elapsed = float(np.median(regions))
    host_cpu_per_wall = float(np.median(host_cpu))
    nms_ties = {"tie_planes_per_batch": round((ties1[0] - ties0[0]) / (NB * n_reg), 2),
                "flood_walk_ms_per_batch": round((ties1[1] - ties0[1]) / (NB * n_reg), 2), "host_threads": ties1[2],
                "note": "planes of a batch whose NMS sibling tie changes the pool: the reference's flood order is walked on a host core for each "
                        "(host ms summed over planes), on the library's process-wide pool of at most host_threads threads"}

    # per-kernel-group GPU times come from runs of their own with the library's per-group events on (str_er_set_profiling): one context alone, and --
    # a short region -- all of them busy; the timed regions run without those events
    serial_prof, _ = rig.serial_profile(d_frames, stages)
    prof_ov = rig.overlapped_profile(d_frames, stages, 2 * P) if world == 1 else {}
    r_tree_stats = rig.filters[0].last_tree_stats()
    r_tile2 = {**rig.filters[0].tile2_stats(), "note": "tiles of the chroma planes given to k_tile_tree2 by this context so far, and how many it handed back to k_tile_tree"}

    # ---- config 3: the OCR scorer on every strong / weak ER of the same batches; then the reference's own call pattern (lines first)
    ocr_leg = group_ocr_leg = None
    if ocr_legs:
        n_o = max(P, NB)            # (as many steps as the headline's region: a shorter region would carry a larger share of the drain of the batches in flight)

        def ocr_leg_run(st, label):
            rig.run(P, d_frames, st)
            els = []
            for _ in range(max(1, args.repeats)):
                el_i, prof_o, last_o = rig.timed(n_o, d_frames, st)
                els.append(el_i)
            el_o = float(np.median(els))
            sp, last1 = rig.serial_profile(d_frames, st)
            out = {"value": round(F * n_o / el_o, 2), "unit": "frames/s", "steps": n_o, "repeats": len(els), "value_min": round(F * n_o / max(els), 2),
                   "value_max": round(F * n_o / min(els), 2), "ms_per_step": round(1e3 * el_o / n_o, 3),
                   "frac_of_value": round(F * n_o / el_o / (F * NB / elapsed), 4), "stages": label}
            return out, sp, last1

        ocr_leg, sp_o, last_o = ocr_leg_run(S.STAGE_ALL | S.STAGE_OCR, "STAGE_ALL | STAGE_OCR: chain_run (slope 0) on every strong / weak ER (src/OCR.cpp:67-140)")
        n_sc = int((last_o.ocr_label >= 0).sum()) if last_o.ocr_label is not None else 0
        k_cls, l_sv, _dim = rig.filters[0].svm_info()
        l_pad, d_q = -(-l_sv // 64) * 64, -(-1800 // 64) * 64
        gemm_ms = sp_o.get("svm_kernel", 0.0)
        forms = rig.filters[0].svm_forms()
        d_q8 = -(-1800 // 128) * 128
        # executed operations of the kernel-matrix launch: one 8-bit multiply-add per (vector, support vector, padded feature) when the model's support vectors are
        # bytes, else three bf16 ones (the f32 value in three pieces)
        flops = 2.0 * n_sc * d_q8 * l_pad if forms["bytes"] else 3 * 2.0 * n_sc * d_q * l_pad
        peak = MFMA_I8_PEAK_TOPS if forms["bytes"] else MFMA_BF16_PEAK_TFLOPS
        tf = flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        alg_flops = 2.0 * n_sc * 1800 * l_sv
        ocr_leg.update({
            "ers_scored_per_batch": n_sc, "svm_model": f"ocr_synth120.model: {k_cls} classes, {l_sv} support vectors, 1800-d, RBF (stand-in for the missing OCR.model, trained on 120 "
                                                       "samples per class like the reference's get_ocr_data, src/utils.cpp:1478-1541)",
            "svm_forms": forms,
            "gpu_ms_per_batch_isolated": {k: round(sp_o.get(k, 0.0), 4) for k in ("ocr_host_gap", "ocr_features", "svm_kernel", "svm_couple")},
            "gpu_ms_note": "ocr_features = k_ocr_list + k_ocr_hist + k_ocr_otsu + k_ocr_features; svm_kernel = the RBF kernel matrix (k_svm_kernel_i8: 8-bit MFMA, exact "
                           "integer distances -- the model's support vectors are 8-bit numerators like the reference's; k_svm_kernel_q, three bf16 MFMAs per product, for any "
                           "other model); svm_couple = k_svm_decide (per-class sums of coefficient x kernel value: f64 MFMA) + k_svm_couple (sigmoid + pairwise coupling); "
                           "ocr_host_gap = stream idle while the host reads the plane counters (the scorer's launch sizes), not GPU work",
            "roofline_svm_kernel": {"bound": "mfma", "kernel": "k_svm_kernel_i8" if forms["bytes"] else "k_svm_kernel_q", "achieved": round(tf, 2), "peak": peak,
                                    "unit": "TOP/s (8-bit)" if forms["bytes"] else "TFLOP/s", "frac": round(tf / peak, 5), "flops_per_launch": int(flops), "avg_launch_ms": round(gemm_ms, 4),
                                    "algorithmic_flops_per_launch": int(alg_flops), "algorithmic_tflops": round(alg_flops / (gemm_ms * 1e-3) / 1e12, 2) if gemm_ms > 0 else 0.0,
                                    "frac_algorithmic": round(alg_flops / (gemm_ms * 1e-3) / 1e12 / peak, 5) if gemm_ms > 0 else 0.0,
                                    "frac_algorithmic_of_bf16_peak": round(alg_flops / (gemm_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 5) if gemm_ms > 0 else 0.0,
                                    "note": (f"2 x N x {d_q8} x l_pad (N = {n_sc} ERs, l_pad = {l_pad}) over the isolated launch: features and support vectors are 8-bit numerators over 255, "
                                             "so x.sv is one v_mfma_i32_32x32x32_i8 per 32 features and |x - sv|^2 an exact integer; " if forms["bytes"] else
                                             f"3 x 2 x N x {d_q} x l_pad (N = {n_sc} ERs, l_pad = {l_pad}) over the isolated launch: the f32 product x.sv as three bf16 MFMAs, f32 accumulate; ") +
                                            "algorithmic = SURVEY 8(d)'s 2 N 1800 l; the launch also evaluates exp() in f64 and writes 8 bytes for every kernel value "
                                            f"({n_sc * l_pad * 8 / 1e6:.0f} MB), which is what bounds it"}})
        if not args.no_cpu_baseline:
            ocr_leg["cpu_baseline"] = cpu_baseline(args.kind, args.workload, cascades, budget_s=8.0, ocr_model=svm_path)
        # the same leg on the small model of rounds 1-5 (319 support vectors, at most 5 a class: k_svm_couple's register build)
        for f in rig.filters:
            f.load_svm_model_text(svm_text_small, 1800)
        small_leg, sp_s, _ = ocr_leg_run(S.STAGE_ALL | S.STAGE_OCR, "the same with the 5-samples-per-class model")
        ocr_leg["small_model_5_per_class"] = {"value": small_leg["value"], "frac_of_value": small_leg["frac_of_value"], "support_vectors": rig.filters[0].svm_info()[1],
                                              "gpu_ms_per_batch_isolated": {k: round(sp_s.get(k, 0.0), 4) for k in ("ocr_host_gap", "ocr_features", "svm_kernel", "svm_couple")}}
        for f in rig.filters:
            f.load_svm_model_text(svm_text, 1800)
        group_ocr_leg, sp_g, last_g = ocr_leg_run(S.STAGE_ALL | st_group | S.STAGE_OCR_LINES,
                                                  "STAGE_ALL | TRACK | GROUP(inner_sup) | OCR_LINES: calc_color, er_track, er_grouping, then chain_run on the members "
                                                  "of the text lines with the line's slope (er_ocr, src/ER.cpp:695-747)")
        group_ocr_leg["line_members_scored_per_batch"] = int(len(last_g.line_label)) if getattr(last_g, "line_label", None) is not None else None
        group_ocr_leg["gpu_ms_per_batch_isolated"] = {k: round(sp_g.get(k, 0.0), 4) for k in ("track", "line_ocr_host_gap", "line_ocr_features", "line_svm_kernel", "line_svm_couple")}

    # ties leg: the same measurement on tie-rich frames (S-ties), so that the cost of exactness is on the line
    ties_leg = None
    if not args.no_ties_leg and args.kind == "text" and world == 1 and args.sibling_order == 0 and not args.ocr and not args.group:
        tf_ = make_frames(S, "ties", W, H, F, first=0, ties_every=args.ties_every)
        d_ties = torch.from_numpy(tf_).to(device)
        torch.cuda.synchronize()
        rig.run(P, d_ties, stages)
        n_t = max(P, NB)
        a0 = rig.tie_totals()
        cpu1 = time.process_time()
        el, _, _ = rig.timed(n_t, d_ties, stages)
        ties_cpu = (time.process_time() - cpu1) / max(el, 1e-9)
        a1 = rig.tie_totals()
        ties_leg = {"value": round(F * n_t / el, 2), "unit": "frames/s", "steps": n_t, "ms_per_step": round(1e3 * el / n_t, 3),
                    "frac_of_value": round(F * n_t / el / (F * NB / elapsed), 4),
                    "tie_planes_per_batch": round((a1[0] - a0[0]) / n_t, 2), "tie_plane_share": round((a1[0] - a0[0]) / n_t / (F * bin(cfg['channel_mask']).count('1') * cfg['n_pyr_levels']), 4),
                    "flood_walk_ms_per_batch": round((a1[1] - a0[1]) / n_t, 2), "host_threads": a1[2], "host_cores": os.cpu_count(), "host_cpu_quota": effective_cpus(), "host_cpus_busy": round(ties_cpu, 2),
                    "note": f"S-ties frames (S-text + one double-L glyph in every {args.ties_every}th frame: an NMS sibling tie with two different outcomes); same "
                            "batches in flight as `value`; flood_walk_ms = host time of the reference-order walks, summed over planes.  The walks are "
                            "bound by host memory latency: once tie planes per batch x walk time / host_threads exceeds the "
                            "GPU's time per batch the leg is host-bound"}
        del d_ties

    # latency leg: ONE frame per call, one batch in flight (north_star: ">= 500 fps end-to-end on 1920x1080" is a
    # per-frame statement; the headline `value` needs 32-frame batches x 6 in flight)
    def latency_leg(w, h, lcfg, frs, d_frs, lstages, with_svm):
        f1 = S.ERFilter(params=S.Params(max_width=w, max_height=h, max_frames=1, n_pyr_levels=lcfg["n_pyr_levels"],
                                        channel_mask=lcfg["channel_mask"], device=dev_index, sibling_order=args.sibling_order))
        f1.load_cascade(0, cascades[0]); f1.load_cascade(1, cascades[1])
        if with_svm:
            f1.load_svm_model_text(svm_text, 1800)
        fb, nF = frs[0].nbytes, len(frs)
        for i in range(3):
            f1.detect_bgr_device(d_frs.data_ptr() + (i % nF) * fb, w, h, 1, lstages)
        n_lat = 40
        dev_ms, host_ms = [], []
        for i in range(n_lat):
            t1 = time.perf_counter()
            f1.detect_bgr_device(d_frs.data_ptr() + (i % nF) * fb, w, h, 1, lstages)
            dev_ms.append(1e3 * (time.perf_counter() - t1))
        for i in range(n_lat):
            t1 = time.perf_counter()
            f1.text_detect(frs[i % nF], lstages)          # pageable host frame: H2D copy inside the call
            host_ms.append(1e3 * (time.perf_counter() - t1))
        f1.close()
        return {"ms_per_frame": round(float(np.median(dev_ms)), 3), "frames_per_s": round(1e3 / float(np.median(dev_ms)), 1),
                "ms_per_frame_p90": round(float(np.percentile(dev_ms, 90)), 3),
                "ms_per_frame_host_input": round(float(np.median(host_ms)), 3),
                "frames_per_s_host_input": round(1e3 / float(np.median(host_ms)), 1),
                "note": f"1 frame per call, 1 call in flight, call-to-return wall time incl. the candidate copy to the host; median of {n_lat}; "
                        "host_input = the frame starts in pageable host memory (H2D inside the call)"}

    latency = None
    rig.close()
    if not args.no_latency and rank == 0:
        latency = latency_leg(W, H, cfg, frames, d_frames, stages, args.ocr)
        if ocr_legs and not args.ocr:
            # ... and with the config-3 scorer behind classify (the host reads the plane counters in between: one more round trip than the plain call)
            lo = latency_leg(W, H, cfg, frames, d_frames, S.STAGE_ALL | S.STAGE_OCR, True)
            latency["with_stage_ocr"] = {k: lo[k] for k in ("ms_per_frame", "frames_per_s", "ms_per_frame_p90")}

    pcie = pcie_nv12 = None
    if not args.no_host_frames and not args.ocr and rank == 0:
        # "a frame = BGR upload excluded and included (both reported)".  The frames sit in the stream's page-locked
        # staging buffers (where a decoder would put them); every step uploads its 3*W*H*F bytes again.
        from concurrent.futures import ThreadPoolExecutor
        st = S.FrameStream(S.Params(max_width=W, max_height=H, max_frames=F, n_pyr_levels=cfg["n_pyr_levels"],
                                    channel_mask=cfg["channel_mask"], device=dev_index, sibling_order=args.sibling_order), depth=P)
        st.load_cascade(0, cascades[0]); st.load_cascade(1, cascades[1])

        def stream_steps(n, data, submit):
            for _ in range(n):
                if st.pending() == P:
                    st.next()
                slot, buf = st.acquire()
                if not filled[slot]:
                    buf[: data.size] = data.reshape(-1)
                    filled[slot] = True
                submit(slot, W, H, F, stages)
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
        stream_steps(max(P, args.warmup), frames, st.submit)
        t1 = time.perf_counter()
        stream_steps(NB, frames, st.submit)
        el = time.perf_counter() - t1
        pcie = {"value": round(F * NB / el, 2), "unit": "frames/s (this rank)", "ms_per_batch": round(1e3 * el / NB, 3),
                "frac_of_value": round(F * NB / el / (F * NB / elapsed), 4),
                "h2d_bytes_per_step": int(frames.size), "h2d_gbs": round(frames.size * NB / el / 1e9, 2),
                # (a torch copy of the same bytes from a torch-pinned tensor, nothing else running: 57 GB/s on most of the round's boxes, 26 on one where the
                # stream itself -- hipHostMalloc'ed staging, several copies queued -- moved 43: the larger of the two is what the link is known to deliver)
                "h2d_gbs_link_alone": round(h2d_gbs, 2), "frames_per_s_at_link_rate": round(max(h2d_gbs, frames.size * NB / el / 1e9) * 1e9 / (frames.size / F), 1),
                "note": "host BGR frames in page-locked memory -> str_er_stream (upload of one batch overlaps the kernels of the others)"}
        # ... and the same frames as a video decoder would deliver them: NV12, half the bytes (build-defined ingest, include/str_er.h)
        with ThreadPoolExecutor(min(F, max(1, (os.cpu_count() or 1) // 2), 16)) as ex:
            nv = np.stack(list(ex.map(S.synth.nv12_from_bgr, frames)))
        filled = [False] * P
        stream_steps(max(P, args.warmup), nv, st.submit_nv12)
        t1 = time.perf_counter()
        stream_steps(NB, nv, st.submit_nv12)
        el = time.perf_counter() - t1
        pcie_nv12 = {"value": round(F * NB / el, 2), "unit": "frames/s (this rank)", "ms_per_batch": round(1e3 * el / NB, 3),
                     "frac_of_value": round(F * NB / el / (F * NB / elapsed), 4),
                     "h2d_bytes_per_step": int(nv.size), "h2d_gbs": round(nv.size * NB / el / 1e9, 2),
                     "note": "the same frames as NV12 (luma + interleaved Cb/Cr at half resolution: what a decoder delivers) through the same stream; "
                             "the NV12 -> Y/Cr/Cb step is build-defined (chroma replicated 2x2), so the planes -- and the candidates -- are not "
                             "those of the BGR frames"}
        st.close()
        del nv

    # ---- config 5 on this one GPU: 3840x2160, 12 levels (the 8-GPU plane / strip sharding of configs[4] is covered by tests; this is the single-GPU rate)
    k4_leg = None
    if single and not args.no_4k_leg and args.size == "1080p" and args.workload == "pyr3x8" and args.kind == "text" and not args.ocr and not args.group:
        del d_frames
        torch.cuda.empty_cache()
        w4, h4, F4, cfg4 = 3840, 2160, 12, WORKLOADS["pyr3x12"]
        fr4 = make_frames(S, "text", w4, h4, F4)
        d4 = torch.from_numpy(fr4).to(device)
        rig4 = Rig(S, P, w4, h4, F4, cfg4, dev_index, args.sibling_order, cascades)
        rig4.run(2 * P, d4, S.STAGE_ALL)
        n4 = max(P, NB)
        el4, prof4, _ = rig4.timed(n4, d4, S.STAGE_ALL)
        sp4, _ = rig4.serial_profile(d4, S.STAGE_ALL)
        ov4 = rig4.overlapped_profile(d4, S.STAGE_ALL, 2 * P)
        px4 = plane_pixels("pyr3x12", w4, h4)
        k4_leg = {"value": round(F4 * n4 / el4, 2), "unit": "frames/s", "steps": n4, "frames_per_step": F4, "ms_per_step": round(1e3 * el4 / n4, 3),
                  "plane_pixels_per_frame": px4, "mpx_per_s": round(px4 * F4 * n4 / el4 / 1e6, 1),
                  "mpx_per_s_of_value": round(plane_pixels(args.workload) * F * NB / elapsed / 1e6, 1),
                  "workload": WORKLOADS["pyr3x12"]["label"] + "; S-text frames; one GPU",
                  "roofline": tile_roofline(px4, F4, sp4.get("tile_tree", 0.0) + sp4.get("tile_tree2", 0.0), ov4.get("tile_tree", 0.0) + ov4.get("tile_tree2", 0.0), "pyr3x12",
                                            sp4.get("tile_tree2", 0.0), 2.0 / 3.0),
                  "gpu_ms_per_step_by_kernel_group_serial": {k: round(v, 4) for k, v in sp4.items()}}
        # the cost of exact NMS ties at this size: the host walk of one 8.3 Mpx plane
        if args.sibling_order == 0 and not args.no_ties_leg:
            t4 = make_frames(S, "ties", w4, h4, F4, ties_every=args.ties_every)
            d4t = torch.from_numpy(t4).to(device)
            rig4.run(P, d4t, S.STAGE_ALL)
            a0 = rig4.tie_totals()
            n4t = max(P, n4 // 2)
            el4t, _, _ = rig4.timed(n4t, d4t, S.STAGE_ALL)
            a1 = rig4.tie_totals()
            walked = a1[0] - a0[0]
            k4_leg["nms_ties"] = {"value": round(F4 * n4t / el4t, 2), "frac_of_4k_value": round(F4 * n4t / el4t / (F4 * n4 / el4), 4),
                                  "tie_planes_per_batch": round(walked / n4t, 2), "flood_walk_ms_per_plane": round((a1[1] - a0[1]) / max(walked, 1), 2)}
            del d4t, t4
        rig4.close()
        if not args.no_latency:
            k4_leg["latency_1frame"] = latency_leg(w4, h4, cfg4, fr4, d4, S.STAGE_ALL, False)
        del d4

    if rank == 0:
        total_frames = F * world * NB
        fps = total_frames / elapsed
        px = plane_pixels(args.workload)
        n_pool = len(r.cands)
        # algorithmic bytes per frame, SURVEY.md 8(d): 3WH (BGR read) + 2*sum(plane px) (each 8-bit
        # plane written once, read once) + N_pool*(2704+48)
        b_alg = 3 * W * H + 2 * px + (n_pool / F) * (2704 + 48)
        # dominant kernel: k_tile_tree reads every plane pixel once -> px bytes per frame.  Its duration is the ISOLATED one
        # (one batch in flight, HIP events on the library's stream around the launch): with P batches sharing the GPU an
        # event-to-event time also contains the other batches' kernels and is not a per-launch cost.
        tile_ms_overlapped = prof_ov.get("tile_tree", 0.0) + prof_ov.get("tile_tree2", 0.0)
        tile_ms = (serial_prof.get("tile_tree", 0.0) + serial_prof.get("tile_tree2", 0.0)) or tile_ms_overlapped
        chans = [i for i in range(6) if cfg["channel_mask"] >> i & 1]
        roof = tile_roofline(px, F, tile_ms, tile_ms_overlapped, args.workload, serial_prof.get("tile_tree2", 0.0), sum(1 for ch in chans if ch % 3) / max(len(chans), 1))
        roof["tile2"] = r_tile2
        roof["path_bytes_per_frame"] = int(b_alg)
        roof["path_frac"] = round(b_alg * fps / world / (HBM_PEAK_GBS * 1e9), 5)
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
            "metric": f"frames/sec (ER extract + 2-stage classify), {W}x{H}",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "repeats": n_reg, "value_min": round(total_frames / max(regions), 2), "value_max": round(total_frames / min(regions), 2),
            "config": {"workload": f"{args.workload}: {cfg['label']}; S-{args.kind} frames" +
                                   ("; + chain-code/SVM OCR scorer on strong+weak ERs (configs[2])" if args.ocr and not args.group else "") +
                                   ("; + calc_color, er_track, er_grouping (text lines)" if args.group else "") +
                                   ("; + chain-code/SVM OCR scorer on the line members with the line slope (er_ocr, configs[2])" if args.ocr and args.group else ""),
                       "frames_per_gpu_per_step": F * inner, "batches_per_step": inner, "frames_per_batch": F,
                       "planes_per_frame": bin(cfg['channel_mask']).count('1') * cfg['n_pyr_levels'],
                       "plane_pixels_per_frame": px, "thresh_step": 8, "min_area": 120, "parallelism": f"frames sharded over {world} GPU(s)",
                       "pooled_per_frame": round(n_pool / F, 1), "batches_in_flight": P, "batch_slots": max(args.batch_slots, 0), "host_cpus_busy": round(host_cpu_per_wall, 2),
                       "workspace_bytes_per_batch_in_flight": ws_bytes, "nms_sibling_ties": "exact (reference flood order)" if args.sibling_order == 0 else "key rule",
                       **({"gather": "RCCL through the C ABI (str_er_gather_last)" if comm is not None else "torch.distributed all_gather"} if world > 1 else {})},
            "nms_ties": nms_ties,
            **({"config3_ocr_leg": ocr_leg} if ocr_leg else {}),
            **({"group_ocr_leg": group_ocr_leg} if group_ocr_leg else {}),
            **({"config5_4k_leg": k4_leg} if k4_leg else {}),
            **({"nms_ties_leg": ties_leg} if ties_leg else {}),
            **({"pcie_inclusive": pcie} if pcie else {}),
            **({"pcie_inclusive_nv12": pcie_nv12} if pcie_nv12 else {}),
            **({"latency_1frame": latency} if latency else {}),
            "roofline": roof,
            "tree_passes_roofline": tree_roof,
            "gpu_ms_per_step_by_kernel_group": {k: round(v, 4) for k, v in prof_ov.items()},
            "gpu_ms_per_step_by_kernel_group_serial": {k: round(v, 4) for k, v in serial_prof.items()},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.kind, args.workload, cascades, ocr_model=svm_path if args.ocr and not args.group else None)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
