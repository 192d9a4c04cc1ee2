#!/usr/bin/env python3
"""VGPRs / scratch bytes / waves per SIMD of every render_kernel instantiation, from ray-tracer_amd/csrc/_obj/resource_usage.txt
(the -Rpass-analysis=kernel-resource-usage remarks of the build).  Template arguments in the kernel's order:
GENERAL MEDIUM TEXTURED LENS COUNT LDSNODES SWAP WIDE LIST RECLDS HALF.   tools/kernel_resources.py [--all] [obj dir]"""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
args = [a for a in sys.argv[1:] if not a.startswith("--")]
obj = Path(args[0]) if args else ROOT / "ray-tracer_amd" / "csrc" / "_obj"
show_all = "--all" in sys.argv
NAMES = "GENERAL MEDIUM TEXTURED LENS COUNT LDSNODES SWAP WIDE LIST RECLDS HALF".split()
txt = (obj / "resource_usage.txt").read_text()
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    if "render_kernel" not in name:
        continue
    m = re.search(r"render_kernelI((?:Lb[01]E|Li\d+E)+)", name)
    vals = re.findall(r"L[bi](\d+)E", m.group(1))
    v = int(re.search(r"VGPRs: (\d+)", b).group(1))
    sc = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
    occ = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1))
    sg = int(re.search(r"SGPRs: (\d+)", b).group(1))
    d = dict(zip(NAMES, vals))
    if not show_all and (d["LENS"] == "1" or d["COUNT"] == "1"):
        continue
    tag = " ".join(f"{k}={d[k]}" for k in NAMES if d.get(k, "0") != "0")
    print(f"{v:4d} VGPRs {sc:5d} B scratch {occ} waves {sg:3d} SGPRs   {tag}")
