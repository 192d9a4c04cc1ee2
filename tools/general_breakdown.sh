#!/bin/bash
# block / cycle breakdown of the three scenes (counting build of the kernels)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for sc in "book_one --width 1200 --height 800 --spp 500" "cornell --width 600 --height 600 --spp 400" "cover --width 800 --height 800 --spp 200"; do
  timeout -k 10 300 python3 $R/bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d['roofline']
print(d['config']['workload'][:14], round(d['value'],1), 'Ms/s kernel_ms', round(r['kernel_ms'],1), 'seg', round(r['segments_per_sample'],2), 'nodes', round(r['nodes_per_sample'],1), 'prims', round(r['prims_per_sample'],1))
print('   util', {k:round(v,2) for k,v in r['simd_utilisation'].items()}, 'cyc', {k:round(v,2) for k,v in r['block_cycle_share'].items()})" || exit 1
done
