#!/bin/bash
# fixed per-launch part (tools/tail_probe.py, depth 100 line) + headline throughput for every library in lib/variants
R=${GRAFT_REPO_ROOT:-$(pwd)}
for f in $R/ray-tracer_amd/lib/variants/librt_*.so; do
  n=$(basename $f .so)
  echo "$n: $(RT_MI355X_LIB=$f timeout -k 10 200 python3 $R/tools/tail_probe.py 2>/dev/null | grep 'depth 100')"
  RT_MI355X_LIB=$f timeout -k 10 200 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('   $n bench', round(d['value'],1), 'Ms/s ms_per_step', round(d['ms_per_step'],2))" || exit 1
done
