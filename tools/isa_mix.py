#!/usr/bin/env python3
"""Static instruction mix of one render_kernel instantiation (from the device assembly hipcc emits for
ray-tracer_amd/csrc/rt_kernels.hip) -> the average issue cost of the VALU instructions the SQ_INSTS_VALU_* class
counters do NOT classify ("other": moves, selects, compares, min/max, lane ops, logic the INT32 counter misses ...).
tools/summarize_profile.py prices the unclassified remainder of SQ_INSTS_VALU with it.

Usage: isa_mix.py <mangled-kernel-substring> [--asm file.s]   (compiles the assembly itself when --asm is absent)"""
import collections
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PRICES = json.load(open(ROOT / "profiles" / "r02_valu_issue.json"))["instructions"]

# classes the PMC counters own (by mnemonic prefix); everything else VALU is "other"
CLASSIFIED = [
    (re.compile(r"v_(add|sub|mul|fma|fmac|mac|mad)_f64|v_(div_fmas|div_fixup|ldexp|fract|frexp|trunc|ceil|floor|rndne|div_scale)_f64"), "f64"),
    (re.compile(r"v_(rcp|rsq|sqrt|trig_preop|log|exp)_f64"), "trans_f64"),
    (re.compile(r"v_(add|sub|subrev|mul|fma|fmac|mac|mad|madak|madmk)_f32"), "f32"),
    (re.compile(r"v_(rcp|rsq|sqrt|log|exp|sin|cos)_f32"), "trans_f32"),
    (re.compile(r"v_cvt_"), "cvt"),
    (re.compile(r"v_(lshlrev|lshrrev|ashrrev|mad_u64|mad_i64|lshl_add)_[bui]64|v_(add|sub)_(co_)?[ui]64|v_lshl_add_u64"), "int64"),
    (re.compile(r"v_(add|sub|subrev|mul_lo|mul_hi|mad|add3|lshl_add|add_lshl|lshl_or|and_or|or3|xad|bfe|bfi|alignbit|lshlrev|lshrrev|ashrrev|and|or|xor|not|bcnt|ffbh|ffbl|mbcnt|addc|subb)_"), "int32"),
]
# issue cost (cycles per wave instruction per SIMD, 4 waves per SIMD) of the "other" mnemonics, from the microbench
OTHER_PRICE = [
    (re.compile(r"v_mov_b32"), PRICES["mov_b32"]["w4"]),
    (re.compile(r"v_mov_b64|v_accvgpr"), 4.2),  # two dwords
    (re.compile(r"v_cndmask_b32.*vcc\s*$"), PRICES["cndmask_b32_sgpr"]["w4"]),  # in real code vcc has just been written: priced like the sgpr form
    (re.compile(r"v_cndmask"), PRICES["cndmask_b32_sgpr"]["w4"]),
    (re.compile(r"v_cmp.*_f64|v_cmp_class_f64"), PRICES["cmp_f64_sgpr"]["w4"]),
    (re.compile(r"v_cmp"), PRICES["cmp_u32_sgpr"]["w4"]),
    (re.compile(r"v_(max|min|max3|min3|med3)_f32"), PRICES["min_f32"]["w4"]),
    (re.compile(r"v_(max|min)_f64"), PRICES["max_f64"]["w4"]),
]
DEFAULT_OTHER = PRICES["cndmask_b32_sgpr"]["w4"]


def nominal(measured):  # pipe occupancy behind a measured issue cost (see tools/summarize_profile.py)
    return 2.0 if measured < 3.2 else (4.0 if measured < 5.5 else (8.0 if measured < 10.0 else 16.0))


def kernel_body(asm, key):
    out, on = [], False
    for line in asm.split("\n"):
        if re.match(r"^_Z\S*render_kernel\S*:", line):
            on = key in line
        if on:
            out.append(line)
            if "s_endpgm" in line:
                break
    return out


def mix(lines):
    counts = collections.Counter()
    other = collections.Counter()
    for line in lines:
        m = re.match(r"^\s+(v_\S+)\s*(.*)$", line)
        if not m:
            continue
        op = m.group(1)
        cls = next((c for rx, c in CLASSIFIED if rx.match(op)), "other")
        counts[cls] += 1
        if cls == "other":
            other[(op, next((p for rx, p in OTHER_PRICE if rx.match(op + " " + m.group(2))), DEFAULT_OTHER))] += 1
    n_other = sum(other.values())
    price = sum(n * nominal(p) for (_, p), n in other.items()) / max(1, n_other)
    top = sorted(((n, op) for (op, _), n in other.items()), reverse=True)[:12]
    return {"static_valu_instructions": dict(counts), "other_price": price, "other_top": [[op, n] for n, op in top]}


def main():
    key = sys.argv[1]
    if "--asm" in sys.argv:
        asm = open(sys.argv[sys.argv.index("--asm") + 1]).read()
    else:
        src = ROOT / "ray-tracer_amd" / "csrc"
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
                            "-mllvm", "-simplifycfg-sink-common=false", "--offload-device-only", "-S", "-o", "-", str(src / "rt_kernels.hip")],
                           capture_output=True, text=True)
        asm = r.stdout
    print(json.dumps(mix(kernel_body(asm, key)), indent=1))


if __name__ == "__main__":
    main()
