#!/bin/bash
# cornell box (general kernel without media / textures) over the libraries in lib/variants
R=${GRAFT_REPO_ROOT:-$(pwd)}
for f in $R/ray-tracer_amd/lib/variants/librt_*.so; do
  n=$(basename $f .so)
  for sc in "cornell --width 600 --height 600 --spp 400" "cover --width 800 --height 800 --spp 200"; do
  RT_MI355X_LIB=$f timeout -k 10 200 python3 $R/bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d['roofline']
print('$n', d['config']['workload'][:10], round(d['value'],1), 'Ms/s kernel_ms', round(r['kernel_ms'],1), 'util', {k:round(v,2) for k,v in r['simd_utilisation'].items()}, 'cyc', {k:round(v,2) for k,v in r['block_cycle_share'].items()})" || exit 1
  done
done
