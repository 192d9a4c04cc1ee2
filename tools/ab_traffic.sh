#!/bin/bash
# GPU box: HBM write traffic + time of render_kernel for the in-tree library and every variant (WRITE_SIZE pass only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for f in $R/ray-tracer_amd/lib/librt_mi355x.so $R/ray-tracer_amd/lib/variants/librt_*.so; do
  [ -f "$f" ] || continue
  n=$(basename $f .so)
  export RT_MI355X_LIB=$f
  rm -rf $R/gpurun_out/traffic_$n
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/traffic_$n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/traffic_$n.log 2>&1
  python3 - <<PY
import csv, glob, collections, json
f=glob.glob("$R/gpurun_out/traffic_$n/*/*_counter_collection.csv")[0]
per=collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if "render_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="WRITE_SIZE":
        per[(r["Dispatch_Id"], r["Kernel_Name"])]+=float(r["Counter_Value"])
vals=[v for (d,k),v in per.items() if ", false, true, true" in k or "false, true, true, false>" in k]
big=[v for v in per.values() if v>1e6]
b=[json.loads(l) for l in open("$R/gpurun_out/traffic_$n.log") if l.startswith("{")][-1]
print("$n", "write GB per launch (full-size dispatches)", [round(v*1024/1e9,2) for v in big][:6], "Ms/s", round(b["value"],1), "kernel_ms", round(b["roofline"]["kernel_ms"],2))
PY
done
