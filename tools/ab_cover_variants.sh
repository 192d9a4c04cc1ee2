#!/bin/bash
# GPU box: the cover (800x800, 300 spp) with the in-tree library and every variant under ray-tracer_amd/lib/variants, interleaved, 2 rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
 for f in $R/ray-tracer_amd/lib/librt_mi355x.so $R/ray-tracer_amd/lib/variants/librt_*.so; do
  [ -f "$f" ] || continue
  RT_MI355X_LIB=$f timeout -k 10 200 python3 $R/tools/cover_lds_probe.py --full --spp 300 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); a=d['default']
print('round $round', '$(basename $f .so)'.ljust(16), 'ms', round(a['kernel_ms_median'],2), a['kernel_ms'][1:], 'lds_nodes', a['lds_nodes'])" || exit 1
 done
done
