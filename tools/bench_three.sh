#!/bin/bash
# GPU box: the bench line of the three BASELINE scenes (default flags = the driver's) + a long book-one run -> gpurun_out/r04_bench_*.log
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python3 $R/bench.py > $R/gpurun_out/r04_bench_book_one.log 2>&1
python3 $R/bench.py --scene cornell --width 600 --height 600 --spp 1000 > $R/gpurun_out/r04_bench_cornell.log 2>&1
python3 $R/bench.py --scene cover --width 800 --height 800 --spp 1000 > $R/gpurun_out/r04_bench_cover.log 2>&1
python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r04_bench_book_one_60.log 2>&1
python3 - <<PY
import json
for s in ("book_one", "cornell", "cover", "book_one_60"):
    d = json.loads([l for l in open("$R/gpurun_out/r04_bench_%s.log" % s) if l.startswith("{")][-1]); r = d["roofline"]
    print(s, round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), "single", round(d["single_render_ms"], 2), round(d["single_render_msamples_per_s"], 1),
          "kernel", round(r["kernel_ms"], 2), "frac", r.get("frac") and round(r["frac"], 3), r.get("frac_envelope_other_at_2_and_4_cycles"),
          "busy", r.get("valu_busy_frac_pmc") and round(r["valu_busy_frac_pmc"], 3), "f64", r.get("f64_math_frac") and round(r["f64_math_frac"], 3),
          "useful", r.get("useful_frac") and round(r["useful_frac"], 3), r.get("reason"), (d.get("cpu_baseline") or {}).get("value"))
PY
