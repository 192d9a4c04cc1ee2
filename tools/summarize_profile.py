#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into profiles-ready files:
summary.json (per-kernel trace stats + per-dispatch PMC means of the render kernel) and kernel_stats.csv."""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

out = Path(sys.argv[1])
tag = sys.argv[2]
dst = out / "summary"
dst.mkdir(exist_ok=True)
res = {"tag": tag, "command": "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (book-one 1200x800x500, depth 100)"}
ks = glob.glob(str(out / "stats" / "*" / "*_kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], dst / "kernel_stats.csv")
    res["kernel_stats"] = [r for r in csv.DictReader(open(ks[0]))][:6]


def is_main(name):  # the timed kernel: COUNT = false (5th template argument)
    if "render_kernel" not in name:
        return False
    args = name[name.index("<") + 1:name.index(">")].replace(" ", "").split(",")
    return args[4] == "false"


pmc = {}
for d in sorted(glob.glob(str(out / "pmc*"))):
    if not Path(d).is_dir():
        continue
    f = glob.glob(d + "/*/*_counter_collection.csv")
    if not f:
        continue
    per = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(f[0])):
        if is_main(r["Kernel_Name"]):
            per[r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
            meta = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
    for c, vals in per.items():
        by_dispatch = collections.defaultdict(float)
        for did, v in vals:
            by_dispatch[did] += v
        pmc[c] = {"mean_per_dispatch": sum(by_dispatch.values()) / len(by_dispatch), "dispatches": len(by_dispatch)}
    res["dispatch_meta"] = meta
res["pmc"] = pmc
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # /opt/skills/guides/MI355X_MICROARCH.md "HBM": counters are KiB; on gfx950 FETCH_SIZE reports 1/2 of the
    # bytes of wide coalesced reads -> doubled (upper bound for this kernel's narrow reads); WRITE_SIZE exact.
    fetch = pmc["FETCH_SIZE"]["mean_per_dispatch"] * 1024 * 2
    write = pmc["WRITE_SIZE"]["mean_per_dispatch"] * 1024
    res["hbm_traffic_bytes_per_launch"] = {"fetch_corrected_x2": fetch, "write": write, "total": fetch + write}
if "SQ_THREAD_CYCLES_VALU" in pmc and "SQ_ACTIVE_INST_VALU" in pmc:
    res["valu_lane_utilisation"] = pmc["SQ_THREAD_CYCLES_VALU"]["mean_per_dispatch"] / (64 * pmc["SQ_ACTIVE_INST_VALU"]["mean_per_dispatch"]) \
        if pmc["SQ_ACTIVE_INST_VALU"]["dispatches"] == pmc["SQ_THREAD_CYCLES_VALU"]["dispatches"] else None
json.dump(res, open(dst / "summary.json", "w"), indent=1)
print(json.dumps({k: res[k] for k in res if k not in ("kernel_stats",)}, indent=1)[:3000])
