#!/usr/bin/env python3
"""Turn gpurun_out/profiles_<tag>/ (tools/collect_profiles.sh) into the tracked summaries under profiles/."""
import collections, csv, json, os, re, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
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
for f in sorted(os.listdir(SRC)):
    if f.startswith("bench_") and f.endswith(".json") and os.path.getsize(os.path.join(SRC, f)) > 0:
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, f))
# 2. kernel stats (rocprofv3 --kernel-trace --stats)
shutil.copy(os.path.join(SRC, "stats", "s_kernel_stats.csv"), os.path.join(DST, f"{tag}_kernel_stats_pyr3x8.csv"))
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
        "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]
with open(os.path.join(DST, f"{tag}_pmc_per_launch_pyr3x8.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(cols)
    for d in rows:
        w.writerow([d.get(c, "") if not isinstance(d.get(c), float) else f"{d[c]:.6g}" for c in cols])
# 4. the number bench.py reads for roofline.traffic
bench = json.load(open(os.path.join(DST, f"bench_{tag}_pyr3x8_text.json")))
F = bench["config"]["frames_per_gpu_per_step"]
tt = {"workload": "pyr3x8", "frames_per_launch": F, "kernel": "k_tile_tree",
      "FETCH_SIZE_KB": fetch["k_tile_tree"]["FETCH_SIZE"], "WRITE_SIZE_KB": write["k_tile_tree"]["WRITE_SIZE"],
      "hbm_bytes_per_launch_raw": (fetch["k_tile_tree"]["FETCH_SIZE"] + write["k_tile_tree"]["WRITE_SIZE"]) * 1024,
      # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> double the read side
      "hbm_bytes_per_launch": (2 * fetch["k_tile_tree"]["FETCH_SIZE"] + write["k_tile_tree"]["WRITE_SIZE"]) * 1024,
      "calibration": {"kernel": "k_bgr_to_ycrcb", "known_read_bytes": 3 * 1920 * 1080 * F, "FETCH_SIZE_bytes": fetch["k_bgr_to_ycrcb"]["FETCH_SIZE"] * 1024,
                      "known_write_bytes": 3 * 1920 * 1080 * F, "WRITE_SIZE_bytes": write["k_bgr_to_ycrcb"]["WRITE_SIZE"] * 1024},
      "note": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE); values are means over the launches of the run; "
              "calibration kernel k_bgr_to_ycrcb (known 3WH*F bytes read and written) confirms FETCH_SIZE = 1/2 of the streamed bytes and "
              "WRITE_SIZE = exact, so hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024; for k_tile_tree's 64-byte row segments the x2 "
              "may over-correct (raw FETCH_SIZE already equals the algorithmic pixel bytes), so the raw sum is kept too"}
json.dump(tt, open(os.path.join(DST, "pmc_tile_tree.json"), "w"), indent=1)
print(json.dumps(tt, indent=1))
