#!/bin/bash
# fixed per-step overhead: bench at small spp
R=${GRAFT_REPO_ROOT:-$(pwd)}
for spp in 1 8 64; do
  python3 $R/bench.py --no-cpu-baseline --spp $spp --steps 20 --warmup 3 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('spp', $spp, 'ms_per_step', round(d['ms_per_step'],3), 'kernel_ms', round(r['kernel_ms'],3))"
done
