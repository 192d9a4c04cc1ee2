#!/bin/bash
# GPU box: the three BASELINE scenes at full size with the in-tree library under each environment in AB_ENVS
# (semicolon-separated lists of VAR=value assignments; an empty item = the default build's behaviour), e.g.
#   AB_ENVS=";RT_BVH_REINSERT=8;RT_BVH_REINSERT=8 RT_BVH_DEPTH_SLACK=4" tools/ab_env.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
IFS=';' read -ra ENVS <<< "${AB_ENVS:-;}"
[ ${#ENVS[@]} -eq 0 ] && ENVS=("")
for e in "${ENVS[@]}"; do
 for sc in "book_one --width 1200 --height 800 --spp 500" "cornell --width 600 --height 600 --spp 1000" "cover --width 800 --height 800 --spp 1000"; do
  env $e timeout -k 10 300 python3 $R/bench.py --scene $sc --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d['roofline']
print('[${e:-default}]', d['config']['workload'][:12], round(d['value'],1), 'Ms/s kernel_ms', round(r['kernel_ms'],2), r['launch']['blocks_per_cu'], r['launch']['lds_bytes'])"
 done
done
