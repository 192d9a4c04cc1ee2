#!/usr/bin/env python3
"""Static VALU instructions of one render_kernel instantiation attributed to source lines (assembly built with
-gline-tables-only).  Usage: isa_by_line.py <mangled-kernel-substring> <file.s> [top]
Prints, per source line, the instruction count by class (f64 / f32 / int / other) -- where the moves, selects and
compares of a kernel come from."""
import collections
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
from isa_mix import CLASSIFIED  # noqa: E402

key, path = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
files = {}
on = False
cur = None
per = collections.defaultdict(collections.Counter)
ops = collections.defaultdict(collections.Counter)
for line in open(path):
    m = re.match(r'^\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', line)
    if m:
        files[int(m.group(1))] = m.group(3)
        continue
    if re.match(r"^_Z\S*render_kernel\S*:", line):
        on = key in line
        continue
    if not on:
        continue
    if "s_endpgm" in line:
        on = False
        continue
    m = re.match(r"^\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"^\s+([vs]_\S+)", line)
    if not m:
        continue
    op = m.group(1)
    if op.startswith("s_"):
        cls = "salu"
    else:
        cls = next((c for rx, c in CLASSIFIED if rx.match(op)), "other")
    per[cur][cls] += 1
    if cls == "other":
        ops[cur][op] += 1
src = {}
def text(f, l):
    if f not in src:
        p = ROOT / "ray-tracer_amd" / "csrc" / f if not f.startswith("rt_rng") else ROOT / "include" / f
        src[f] = open(p).read().split("\n") if p.exists() else []
    return src[f][l - 1].strip()[:110] if 0 < l <= len(src[f]) else ""
tot = collections.Counter()
for c in per.values():
    tot.update(c)
print("total", dict(tot))
rows = sorted(per.items(), key=lambda kv: -kv[1]["other"])[:top]
for (f, l), c in rows:
    print(f"{f}:{l:4d} other {c['other']:4d} f64 {c['f64']:4d} int {c['int32'] + c['int64']:4d} salu {c['salu']:4d} | {dict(ops[(f, l)].most_common(4))} | {text(f, l)}")
