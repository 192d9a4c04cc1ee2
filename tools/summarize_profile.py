#!/usr/bin/env python3
"""Condense a tools/profile_scene.sh output directory into profiles-ready files:
summary/summary.json + summary/kernel_stats.csv.

summary.json holds, for the timed render_kernel (COUNT = false instantiation):
  * the rocprofv3 --kernel-trace --stats rows,
  * per-dispatch means of every PMC counter collected (separate --pmc passes),
  * HBM bytes per launch (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE; KiB -> bytes),
  * the VALU-issue roofline: sum over instruction classes of (wave-instruction count x issue cycles per
    instruction per SIMD, measured by tools/microbench/valu_issue.hip -> profiles/r02_valu_issue.json)
    divided by (1024 SIMDs x elapsed shader cycles of the launch, GRBM_GUI_ACTIVE / 8 XCDs),
  * register / scratch / LDS figures of the code object (hipcc -Rpass-analysis=kernel-resource-usage remarks kept in
    ray-tracer_amd/csrc/_obj/resource_usage.txt, matched by the demangled kernel name, c++filt) -- rocprofv3's own
    VGPR_Count / LDS_Block_Size columns are wrong for this launch (64 / 0) and are reported only as "rocprof_dispatch_columns".

Usage: summarize_profile.py <prof dir> <tag> [--prices profiles/r02_valu_issue.json]
"""
import collections
import csv
import glob
import json
import re
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
N_SIMD = 256 * 4  # MI355X: 256 CUs x 4 SIMDs (MI355X_MICROARCH.md "Chip-level parameters")

# issue cycles per wave64 instruction per SIMD; overwritten by the measured table when present
DEFAULT_PRICES = {"f32": 2.0, "f64": 4.0, "trans_f32": 8.0, "trans_f64": 16.0, "int32": 2.0, "int64": 4.0, "cvt": 4.0, "other": 2.0,
                  "source": "defaults from MI355X_MICROARCH.md cycle constants (v_fma_f32 2 cyc/SIMD, f64 half rate); NOT measured"}


def nominal(measured):
    """pipe occupancy behind a measured issue cost: the microbenchmark sees 2.3-2.4 / 4.15-4.35 / 6.3-8.3 / 16.2 cycles (the
    pipe's 2 / 4 / 8 / 16 plus a few per cent of issue arbitration); the roofline charges the nominal figure, so that a
    kernel cannot come out above 1 through prices that were rounded up"""
    return 2.0 if measured < 3.2 else (4.0 if measured < 5.5 else (8.0 if measured < 10.0 else 16.0))


def load_prices(path, waves, other_price):
    """issue cycles per wave instruction per SIMD of each counter class at `waves` waves per SIMD"""
    p = dict(DEFAULT_PRICES)
    if path and Path(path).exists():
        m = json.load(open(path))
        ins = m["instructions"]
        col = "w%d" % max(1, min(4, int(waves)))
        w = lambda n: nominal(ins[n][col])  # noqa: E731
        mean = lambda *ns: sum(w(n) for n in ns) / len(ns)  # noqa: E731
        p = {
            "f32": mean("fma_f32", "add_f32", "mul_f32"),
            "f64": mean("fma_f64", "add_f64", "mul_f64"),
            "trans_f32": w("rcp_f32"),
            "trans_f64": mean("rcp_f64", "rsq_f64", "sqrt_f64"),
            # the INT32 counter mixes full-rate (add, and / or / xor, right shifts) and half-rate (left shifts, multiplies,
            # bit-field, carry) instructions: the mean of the two groups
            "int32": (mean("add_u32", "xor_b32", "and_b32", "lshrrev_b32") + mean("lshlrev_b32", "mul_lo_u32", "bfe_u32", "add_co_u32")) / 2,
            "int64": mean("lshlrev_b64", "lshrrev_b64", "mad_u64_u32"),
            "cvt": mean("cvt_f64_u32", "cvt_f32_f64", "cvt_f64_f32", "cvt_u32_f64"),
            "other": other_price if other_price else mean("mov_b32", "cndmask_b32_sgpr", "cmp_f64_sgpr"),
            "source": f"{Path(path).name}: measured on the MI355X by tools/microbench/valu_issue.hip, column {col} "
                      f"({waves} waves per SIMD), each instruction charged its pipe occupancy 2 / 4 / 8 / 16; 'other' = static-mix average of the unclassified VALU instructions of this kernel "
                      f"(tools/isa_mix.py)" + ("" if other_price else " -- unavailable, mean of mov / cndmask / cmp used"),
        }
    return p


def kernel_template_args(name):
    return name[name.index("<") + 1:name.index(">")].replace(" ", "").split(",")


def is_main(name):  # the timed kernel: COUNT = false (5th template argument)
    return "render_kernel" in name and kernel_template_args(name)[4] == "false"


def code_object_info(demangled_name):
    """registers / scratch / static LDS of the kernel from the compiler's resource-usage remarks"""
    ru = ROOT / "ray-tracer_amd" / "csrc" / "_obj" / "resource_usage.txt"
    if not ru.exists():
        return None
    blocks, cur = {}, None
    for line in open(ru):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            blocks[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\S+) \[-Rpass", line)
        if m and cur:
            blocks[cur][m.group(1).strip()] = m.group(2)
    try:
        dem = subprocess.run(["c++filt"] + list(blocks), capture_output=True, text=True).stdout.split("\n")
    except OSError:
        return None
    want = demangled_name.replace(" ", "")
    for mangled, d in zip(blocks, dem):
        if d.replace(" ", "") == want:
            b = blocks[mangled]
            return {"vgprs": int(b.get("VGPRs", -1)), "agprs": int(b.get("AGPRs", -1)), "sgprs": int(b.get("TotalSGPRs", -1)),
                    "scratch_bytes_per_lane": int(b.get("ScratchSize [bytes/lane]", -1)),
                    "occupancy_waves_per_simd": int(b.get("Occupancy [waves/SIMD]", -1)),
                    "static_lds_bytes": int(b.get("LDS Size [bytes/block]", -1)), "mangled": mangled,
                    "source": "hipcc -Rpass-analysis=kernel-resource-usage (ray-tracer_amd/csrc/_obj/resource_usage.txt)"}
    return None


def main():
    out = Path(sys.argv[1])
    tag = sys.argv[2]
    prices_path = ROOT / "profiles" / "r02_valu_issue.json"
    if "--prices" in sys.argv:
        prices_path = Path(sys.argv[sys.argv.index("--prices") + 1])
    dst = out / "summary"
    dst.mkdir(exist_ok=True)
    cmd = (out / "command.txt").read_text().strip() if (out / "command.txt").exists() else \
        "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (book-one 1200x800x500, depth 100)"
    res = {"tag": tag, "command": cmd}
    # the bench line of the stats pass (launch configuration as the library reports it)
    sl = out / "stats.log"
    if sl.exists():
        for line in open(sl):
            if line.startswith("{") and '"metric"' in line:
                b = json.loads(line)
                res["bench_line_of_stats_pass"] = {k: b.get(k) for k in ("value", "unit", "ms_per_step", "config")}
                res["launch"] = (b.get("roofline") or {}).get("launch")
                res["library"] = (b.get("config") or {}).get("library")
    newest = lambda files: sorted(files, key=lambda f: Path(f).stat().st_mtime)[-1:]  # noqa: E731  (merged reruns leave older files behind)
    ks = newest(glob.glob(str(out / "stats" / "*" / "*_kernel_stats.csv")))
    main_name = None
    if ks:
        shutil.copy(ks[0], dst / "kernel_stats.csv")
        rows = [r for r in csv.DictReader(open(ks[0]))]
        res["kernel_stats"] = rows[:6]
        for r in rows:
            if is_main(r["Name"]):
                main_name = r["Name"]
                res["render_kernel_avg_ms"] = float(r["AverageNs"]) / 1e6
                res["render_kernel_calls"] = int(r["Calls"])

    pmc = {}
    for d in sorted(glob.glob(str(out / "pmc*"))):
        if not Path(d).is_dir():
            continue
        f = newest(glob.glob(d + "/*/*_counter_collection.csv"))
        if not f:
            continue
        per = collections.defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(f[0])):
            if is_main(r["Kernel_Name"]):
                main_name = main_name or r["Kernel_Name"]
                per[r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
                meta = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
        for c, vals in per.items():
            by_dispatch = collections.defaultdict(float)
            for did, v in vals:
                by_dispatch[did] += v
            pmc[c] = {"mean_per_dispatch": sum(by_dispatch.values()) / len(by_dispatch), "dispatches": len(by_dispatch)}
        if meta:
            res["rocprof_dispatch_columns"] = dict(meta, note="as rocprofv3 prints them; VGPR / LDS columns are NOT the code object's (see code_object)")
    res["pmc"] = pmc
    if main_name:
        res["kernel"] = main_name
        res["code_object"] = code_object_info(main_name)

    def mean(c):
        return pmc[c]["mean_per_dispatch"] if c in pmc else None

    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        # /opt/skills/guides/MI355X_MICROARCH.md "HBM": counters are KiB; on gfx950 FETCH_SIZE reports 1/2 of the
        # bytes of wide coalesced reads -> doubled (upper bound for this kernel's narrow reads); WRITE_SIZE exact.
        fetch = mean("FETCH_SIZE") * 1024 * 2
        write = mean("WRITE_SIZE") * 1024
        res["hbm_traffic_bytes_per_launch"] = {"fetch_corrected_x2": fetch, "write": write, "total": fetch + write}
    if "SQ_THREAD_CYCLES_VALU" in pmc and "SQ_ACTIVE_INST_VALU" in pmc:
        res["valu_lane_utilisation"] = mean("SQ_THREAD_CYCLES_VALU") / (64 * mean("SQ_ACTIVE_INST_VALU"))

    # ---- VALU-issue roofline ----
    classes = {"f64": ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"], "trans_f64": ["SQ_INSTS_VALU_TRANS_F64"],
               "f32": ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32"], "trans_f32": ["SQ_INSTS_VALU_TRANS_F32"],
               "int32": ["SQ_INSTS_VALU_INT32"], "int64": ["SQ_INSTS_VALU_INT64"], "cvt": ["SQ_INSTS_VALU_CVT"]}
    need = [c for cs in classes.values() for c in cs] + ["SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"]
    if all(c in pmc for c in need):
        waves = (res.get("launch") or {}).get("waves_per_simd") or 4
        other_price, other_mix = None, None
        try:
            sys.path.insert(0, str(ROOT / "tools"))
            import isa_mix
            asm_path = Path("/tmp/rt_kernels_for_profile.s")
            if not asm_path.exists():
                src = ROOT / "ray-tracer_amd" / "csrc"
                subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
                                "-mllvm", "-simplifycfg-sink-common=false", "--offload-device-only", "-S", "-o", str(asm_path),
                                str(src / "rt_kernels.hip")], capture_output=True)
            key = (res.get("code_object") or {}).get("mangled", "").split("render_kernel")[-1]
            other_mix = isa_mix.mix(isa_mix.kernel_body(asm_path.read_text(), key))
            other_price = other_mix["other_price"]
        except Exception as e:  # noqa: BLE001
            other_mix = {"error": repr(e)}
        prices = load_prices(prices_path, waves, other_price)
        res["static_isa_mix"] = other_mix
        counts = {k: sum(mean(c) for c in cs) for k, cs in classes.items()}
        counts["other"] = max(0.0, mean("SQ_INSTS_VALU") - sum(counts.values()))
        issue = {k: counts[k] * prices[k] for k in counts}
        elapsed_cycles = mean("GRBM_GUI_ACTIVE") / 8.0  # the counter sums the 8 XCDs
        total_issue = sum(issue.values())
        frac = total_issue / (N_SIMD * elapsed_cycles)
        lane = res.get("valu_lane_utilisation")
        n_other, rest = counts["other"], total_issue - issue["other"]
        busy = mean("SQ_ACTIVE_INST_VALU")
        res["valu_issue_roofline"] = {
            "bound": "valu_issue",
            # no hardware counter classifies moves / selects / compares / lane operations: the envelope prices the unclassified
            # instructions at the cheapest (2) and the dearest (4 cycles) of them instead of the static-mix average
            "frac_envelope_other_at_2_and_4_cycles": [(rest + 2.0 * n_other) / (N_SIMD * elapsed_cycles), (rest + 4.0 * n_other) / (N_SIMD * elapsed_cycles)],
            "f64_math_frac": (issue["f64"] + issue["trans_f64"]) / (N_SIMD * elapsed_cycles),
            # quad-cycles with a VALU instruction in flight x 4 over the SIMD-cycles of the launch: the hardware's own measure
            "valu_busy_frac_pmc": (busy * 4.0 / (N_SIMD * elapsed_cycles)) if busy else None,
            "wave_instructions_per_launch": counts, "issue_cycles_per_instruction": {k: prices[k] for k in counts},
            "prices_source": prices["source"],
            "issue_cycles_per_launch": issue, "issue_cycles_total": total_issue,
            "elapsed_shader_cycles": elapsed_cycles, "n_simd": N_SIMD,
            "peak_issue_cycles": N_SIMD * elapsed_cycles,
            "frac": frac,
            "useful_frac": frac * lane if lane else None,
            "note": "frac = sum(class count x issue cycles) / (1024 SIMDs x elapsed cycles): share of the VALU issue slots of the whole "
                    "chip the launch filled; useful_frac weights it with the VALU lane utilisation (SQ_THREAD_CYCLES_VALU / 64 / "
                    "SQ_ACTIVE_INST_VALU). 'other' = SQ_INSTS_VALU minus the classified counts (moves, selects, compares, min / max, "
                    "lane ops), priced with the static-mix average of those instructions in this kernel's code object",
        }
        if res.get("render_kernel_avg_ms"):
            res["valu_issue_roofline"]["clock_ghz"] = elapsed_cycles / (res["render_kernel_avg_ms"] * 1e-3) / 1e9
    if "SQ_LDS_BANK_CONFLICT" in pmc and "SQ_ACTIVE_INST_LDS" in pmc:
        res["lds_conflict_share_of_lds_active"] = mean("SQ_LDS_BANK_CONFLICT") / mean("SQ_ACTIVE_INST_LDS")
    if "SQ_WAVE_CYCLES" in pmc:
        wc = mean("SQ_WAVE_CYCLES")
        res["wave_cycle_shares"] = {c: mean(c) / wc for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if c in pmc}
    json.dump(res, open(dst / "summary.json", "w"), indent=1)
    brief = {k: res[k] for k in res if k not in ("kernel_stats", "pmc")}
    print(json.dumps(brief, indent=1)[:6000])


if __name__ == "__main__":
    main()
