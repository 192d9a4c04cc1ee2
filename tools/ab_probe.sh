#!/bin/bash
# swap_cycles share for each RT_SWAP_PROBE variant under ray-tracer_amd/lib/variants (book-one)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for f in $R/ray-tracer_amd/lib/variants/librt_*.so; do
  n=$(basename $f .so)
  RT_MI355X_LIB=$f timeout -k 10 200 python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d['roofline']
print('$n', round(d['value'],1), 'swap share', round(r['block_cycle_share']['swap'],4), 'finish', round(r['block_cycle_share']['finish'],3))" || exit 1
done
