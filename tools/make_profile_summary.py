#!/usr/bin/env python3
"""Turn gpurun_out/profiles_<tag>/ (tools/collect_profiles.sh) into the tracked summaries under profiles/."""
import collections, csv, json, os, re, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "r03"
PMC_JSON_ONLY = "--pmc-json-only" in sys.argv      # (on the GPU box, between the counter passes and the bench lines: tools/collect_profiles.sh)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", f"profiles_{tag}")
DST = os.path.join(ROOT, "profiles")
os.makedirs(DST, exist_ok=True)


def short(n):
    m = re.match(r"(?:void )?(?:str_er::)?([A-Za-z_0-9]+)", n)     # "void str_er::k_tile_tree<480>(...)" -> k_tile_tree
    return m.group(1) if m else n


def pmc(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


# 1. bench lines
for f in ([] if PMC_JSON_ONLY else sorted(os.listdir(SRC))):
    if f.startswith("bench_") and f.endswith(".json") and os.path.getsize(os.path.join(SRC, f)) > 0:
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, f))
# 2. kernel stats (rocprofv3 --kernel-trace --stats)
if not PMC_JSON_ONLY:
    shutil.copy(os.path.join(SRC, "stats", "s_kernel_stats.csv"), os.path.join(DST, f"{tag}_kernel_stats_pyr3x8.csv"))
    for extra, name in (("stats_config3", "kernel_stats_config3"), ("stats_group_ocr", "kernel_stats_group_ocr"), ("stats_4k", "kernel_stats_4k")):
        f = os.path.join(SRC, extra, "s_kernel_stats.csv")
        if os.path.exists(f):
            shutil.copy(f, os.path.join(DST, f"{tag}_{name}.csv"))
    for extra, name in ((os.path.join("pmc_config3", "summary.txt"), "pmc_config3.txt"), (os.path.join("latency", "timeline.txt"), "latency_1frame_timeline.txt"),
                        (os.path.join("latency", "latency.txt"), "latency_1frame_stages.txt")):
        f = os.path.join(SRC, extra)
        if os.path.exists(f):
            shutil.copy(f, os.path.join(DST, f"{tag}_{name}"))
# 3. PMC summaries
fetch = pmc(os.path.join(SRC, "pmc_fetch", "p_counter_collection.csv"))
write = pmc(os.path.join(SRC, "pmc_write", "p_counter_collection.csv"))
sq = pmc(os.path.join(SRC, "pmc_sq1", "p_counter_collection.csv"))
sq2 = pmc(os.path.join(SRC, "pmc_sq2", "p_counter_collection.csv"))
rows = []
for k in sorted(fetch, key=lambda k: -fetch[k].get("FETCH_SIZE", 0)):
    if not k.startswith("k_"):
        continue
    d = {"kernel": k, "FETCH_SIZE_KB": fetch[k].get("FETCH_SIZE"), "WRITE_SIZE_KB": write.get(k, {}).get("WRITE_SIZE")}
    d.update(sq.get(k, {}))
    d.update(sq2.get(k, {}))
    rows.append(d)
cols = ["kernel", "FETCH_SIZE_KB", "WRITE_SIZE_KB", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU",
        "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS",
        "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_VMEM_RD", "SQ_INSTS_BRANCH"]
with open(os.path.join(DST, f"{tag}_pmc_per_launch_pyr3x8.csv") if not PMC_JSON_ONLY else os.devnull, "w") as f:
    w = csv.writer(f)
    w.writerow(cols)
    for d in rows:
        w.writerow([d.get(c, "") if not isinstance(d.get(c), float) else f"{d[c]:.6g}" for c in cols])
# 4. the number bench.py reads for roofline.traffic
_cfg = {} if PMC_JSON_ONLY else json.load(open(os.path.join(DST, f"bench_{tag}_pyr3x8_text.json")))["config"]
F = 32 if PMC_JSON_ONLY else _cfg.get("frames_per_batch", _cfg["frames_per_gpu_per_step"])      # (json-only: bench.py's default batch on pyr3x8, which the counter passes ran)
# (round 6: the tile trees of a batch are built by two kernels -- k_tile_tree2 on the chroma planes, k_tile_tree on the luma planes, k_tile_tree_fb on the
# tiles the first hands back: their traffic is added up, like their times in bench.py's roofline)
TILE_KERNELS = [k for k in ("k_tile_tree2", "k_tile_tree", "k_tile_tree_fb") if k in fetch]
t_fetch = sum(fetch[k]["FETCH_SIZE"] for k in TILE_KERNELS)
t_write = sum(write.get(k, {}).get("WRITE_SIZE", 0.0) for k in TILE_KERNELS)
tt = {"workload": "pyr3x8", "frames_per_launch": F, "kernel": " + ".join(TILE_KERNELS),
      "per_kernel_KB": {k: {"FETCH_SIZE": fetch[k]["FETCH_SIZE"], "WRITE_SIZE": write.get(k, {}).get("WRITE_SIZE")} for k in TILE_KERNELS},
      "FETCH_SIZE_KB": t_fetch, "WRITE_SIZE_KB": t_write,
      "hbm_bytes_per_launch_raw": (t_fetch + t_write) * 1024,
      # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> double the read side
      "hbm_bytes_per_launch": (2 * t_fetch + t_write) * 1024,
      "calibration": {"kernel": "k_bgr_to_ycrcb", "known_read_bytes": 3 * 1920 * 1080 * F, "FETCH_SIZE_bytes": fetch["k_bgr_to_ycrcb"]["FETCH_SIZE"] * 1024,
                      "known_write_bytes": 3 * 1920 * 1080 * F, "WRITE_SIZE_bytes": write["k_bgr_to_ycrcb"]["WRITE_SIZE"] * 1024},
      "note": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE); values are means over the launches of the run; "
              "calibration kernel k_bgr_to_ycrcb (known 3WH*F bytes read and written) confirms FETCH_SIZE = 1/2 of the streamed bytes and "
              "WRITE_SIZE = exact, so hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024; for k_tile_tree's 64-byte row segments the x2 "
              "may over-correct (raw FETCH_SIZE already equals the algorithmic pixel bytes), so the raw sum is kept too"}
json.dump(tt, open(os.path.join(DST, "pmc_tile_tree.json"), "w"), indent=1)
print(json.dumps(tt, indent=1))

# 5. the numbers of the round as one table (profiles/<tag>_numbers.md), so that README.md can stay prose
if not PMC_JSON_ONLY:
    lines = [f"# Round {tag[1:].lstrip('0')} numbers (generated by tools/make_profile_summary.py from the files beside it)", "",
             "| bench line | frames/s | ms per step (a step = `config.batches_per_step` batches) | tile trees (`k_tile_tree2` + `k_tile_tree`) ms per batch (isolated) | GB/s (algorithmic plane bytes) | frac of 8 TB/s | 1-frame latency ms | PCIe-inclusive frames/s | CPU baseline frames/s |",
             "|---|---|---|---|---|---|---|---|---|"]
    for n in ("pyr3x8_text", "native6_text", "native6_noise"):
        fn = os.path.join(DST, f"bench_{tag}_{n}.json")
        if not os.path.exists(fn):
            continue
        d = json.load(open(fn))
        r = d["roofline"]
        lines.append(f"| `bench_{tag}_{n}.json` | **{d['value']:.0f}** | {d['ms_per_step']:.2f} | {r['avg_launch_ms']:.3f} | {r['achieved']:.1f} | {100 * r['frac']:.2f} % | "
                     f"{(d.get('latency_1frame') or {}).get('ms_per_frame', '-')} | {(d.get('pcie_inclusive') or {}).get('value', '-')} | {(d.get('cpu_baseline') or {}).get('value', '-')} |")
    d = json.load(open(os.path.join(DST, f"bench_{tag}_pyr3x8_text.json")))
    lines += ["", f"Per-kernel GPU time per {F}-frame batch of pyr3x8, one batch in flight (HIP events, ms): " +
              ", ".join(f"{k} {v:.2f}" for k, v in d["gpu_ms_per_step_by_kernel_group_serial"].items()) +
              f" (sum {sum(d['gpu_ms_per_step_by_kernel_group_serial'].values()):.2f}; the timed region, six batches overlapping, needs {d['ms_per_step'] / d['config'].get('batches_per_step', 1):.2f} ms per batch).", ""]
    st = {short(r["Name"]): r for r in csv.DictReader(open(os.path.join(DST, f"{tag}_kernel_stats_pyr3x8.csv")))}
    tks = [k for k in ("k_tile_tree2", "k_tile_tree", "k_tile_tree_fb") if k in st]
    tt_ms = sum(float(st[k]["AverageNs"]) for k in tks) / 1e6
    alg = 12395367 * F
    lines += [f"`{tag}_kernel_stats_pyr3x8.csv` (rocprofv3, one batch in flight): " + ", ".join(f"`{k}` {st[k]['Calls']} calls, average {float(st[k]['AverageNs']) / 1e6:.3f} ms" for k in tks) +
              f" -> the tile trees of a batch {tt_ms:.3f} ms -> {alg} B / {tt_ms:.3f} ms = {alg / tt_ms / 1e6:.1f} GB/s = {alg / tt_ms / 1e6 / 8000:.4f} of 8 TB/s (the bench line's "
              f"HIP-event figure: {d['roofline']['avg_launch_ms']:.3f} ms, frac {d['roofline']['frac']:.4f}; per kernel: {d['roofline'].get('per_kernel')}).", ""]
    pm = {r["kernel"]: r for r in csv.DictReader(open(os.path.join(DST, f"{tag}_pmc_per_launch_pyr3x8.csv")))}
    for tk in [k for k in ("k_tile_tree2",) if k in pm]:
        t2 = pm[tk]
        w2 = float(t2["SQ_WAVES"])
        lines += [f"`{tk}` counters per launch of {F} frames ({w2:.0f} waves = pairs of tiles): FETCH_SIZE {float(t2['FETCH_SIZE_KB']) / 1e6:.3f} GB raw, WRITE_SIZE {float(t2['WRITE_SIZE_KB']) / 1e6:.3f} GB; "
                  f"per wave {float(t2['SQ_INSTS_VALU']) / w2:.0f} vector + {float(t2['SQ_INSTS_SALU']) / w2:.0f} scalar + {float(t2['SQ_INSTS_LDS']) / w2:.0f} LDS instructions "
                  f"(= {float(t2['SQ_INSTS_VALU']) / w2 / 8:.0f} vector instructions per 512 pixels; `k_tile_tree`: its per-wave figure below).", ""]
    t = pm["k_tile_tree"]
    w = float(t["SQ_WAVES"])
    lines += [f"`k_tile_tree` counters per launch of {F} frames: FETCH_SIZE {float(t['FETCH_SIZE_KB']) / 1e6:.3f} GB raw (algorithmic plane bytes {alg / 1e9:.3f} GB), WRITE_SIZE {float(t['WRITE_SIZE_KB']) / 1e6:.3f} GB; "
              f"per wave {float(t['SQ_INSTS_VALU']) / w:.0f} vector + {float(t['SQ_INSTS_SALU']) / w:.0f} scalar + {float(t['SQ_INSTS_LDS']) / w:.0f} LDS"
              + (f" + {float(t['SQ_INSTS_BRANCH']) / w:.0f} branch" if t.get('SQ_INSTS_BRANCH') else "") + " instructions; "
              f"of the cycles its waves are resident {100 * float(t['SQ_ACTIVE_INST_ANY']) / float(t['SQ_WAVE_CYCLES']):.1f} % issue an instruction "
              f"(vector {100 * float(t['SQ_ACTIVE_INST_VALU']) / float(t['SQ_WAVE_CYCLES']):.1f} %), {100 * float(t['SQ_WAIT_ANY']) / float(t['SQ_WAVE_CYCLES']):.0f} % are parked "
              f"(s_waitcnt / barrier) and {100 * float(t['SQ_WAIT_INST_ANY']) / float(t['SQ_WAVE_CYCLES']):.0f} % wait for an issue slot; "
              f"LDS bank conflict cycles / LDS busy cycles = {100 * float(t['SQ_LDS_BANK_CONFLICT']) / float(t['SQ_ACTIVE_INST_LDS']):.0f} %.", ""]
    # the legs of the default line (round 5)
    def leg(k, fmt):
        v = d.get(k)
        if v:
            lines.append(fmt(v))
    leg("config3_ocr_leg", lambda v: f"`config3_ocr_leg` (BASELINE configs[2]: chain-code + SVM scorer on the {v['ers_scored_per_batch']} strong / weak ERs of a batch): **{v['value']:.0f}** frames/s = "
        f"{v['frac_of_value']:.3f} of `value`; isolated GPU ms per batch {v['gpu_ms_per_batch_isolated']}; `{v['roofline_svm_kernel']['kernel']}` {v['roofline_svm_kernel']['achieved']} {v['roofline_svm_kernel']['unit']} = "
        f"{v['roofline_svm_kernel']['frac']:.3f} of the {v['roofline_svm_kernel']['peak']} peak (algorithmic 2 N 1800 l: {v['roofline_svm_kernel'].get('algorithmic_tflops')} = {v['roofline_svm_kernel'].get('frac_algorithmic')}); "
        f"the 5-per-class model: {v.get('small_model_5_per_class')}; CPU baseline {(v.get('cpu_baseline') or {}).get('value', '-')} frames/s.")
    leg("group_ocr_leg", lambda v: f"`group_ocr_leg` (calc_color, er_track, er_grouping, then the scorer on the {v['line_members_scored_per_batch']} line members of a batch): **{v['value']:.0f}** frames/s = "
        f"{v['frac_of_value']:.3f} of `value`; isolated GPU ms per batch {v['gpu_ms_per_batch_isolated']}.")
    leg("config5_4k_leg", lambda v: f"`config5_4k_leg` (BASELINE configs[4] on one GPU, 3840x2160 x 12 levels, {v['frames_per_step']} frames per batch): **{v['value']:.0f}** frames/s ({v['mpx_per_s']:.0f} Mpx/s; the 1080p line: "
        f"{v['mpx_per_s_of_value']:.0f}); tile trees {v['roofline']['avg_launch_ms']} ms per batch = {v['roofline']['frac']:.4f} of 8 TB/s; 1-frame latency "
        f"{(v.get('latency_1frame') or {}).get('ms_per_frame', '-')} ms; exact NMS ties: {v.get('nms_ties')}.")
    leg("nms_ties_leg", lambda v: f"`nms_ties_leg`: **{v['value']:.0f}** frames/s = {v['frac_of_value']:.3f} of `value` at {v['tie_planes_per_batch']} tie planes per batch ({v['flood_walk_ms_per_batch']} ms of host walks, {v['host_threads']} threads).")
    leg("pcie_inclusive", lambda v: f"`pcie_inclusive`: **{v['value']:.0f}** frames/s = {v['frac_of_value']:.3f} of `value` ({v['h2d_gbs']} GB/s; the link alone {v['h2d_gbs_link_alone']} GB/s); NV12: "
        f"{(d.get('pcie_inclusive_nv12') or {}).get('value', '-')} ({(d.get('pcie_inclusive_nv12') or {}).get('frac_of_value', '-')}).")
    lines.append(f"`value` = median of {d.get('repeats', 1)} timed regions: min {d.get('value_min', '-')}, max {d.get('value_max', '-')}.")
    lines.append("")
    open(os.path.join(DST, f"{tag}_numbers.md"), "w").write("\n".join(lines))
    print("\n".join(lines))
