#!/bin/bash
# GPU box: the three BASELINE scenes at full size with the in-tree library and every variant under ray-tracer_amd/lib/variants
R=${GRAFT_REPO_ROOT:-$(pwd)}
for f in $R/ray-tracer_amd/lib/librt_mi355x.so $R/ray-tracer_amd/lib/variants/librt_*.so; do
 [ -f "$f" ] || continue
 for sc in "cornell --width 600 --height 600 --spp 1000" "cover --width 800 --height 800 --spp 1000"; do
  RT_MI355X_LIB=$f timeout -k 10 300 python3 $R/bench.py --scene $sc --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d['roofline']
print('$(basename $f .so)', d['config']['workload'][:12], round(d['value'],1), 'Ms/s kernel_ms', round(r['kernel_ms'],1), {k:round(v,2) for k,v in r['simd_utilisation'].items()}, {k:round(v,2) for k,v in r['block_cycle_share'].items()}, r['launch']['blocks_per_cu'], r['launch']['lds_bytes'])"
 done
done
