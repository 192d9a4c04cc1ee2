#!/bin/bash
# GPU box: where the swap-at-shade bookkeeping's cycles go, step by step.  Variant libraries librt_probe<k>.so (tools/build_part1_variant.sh
# probe<k> "-DRT_SWAP_PROBE=<k>") stop the counting build's swap clock after step k of the shade block's swap (1 classify + settle, 2 mode,
# 3 locks, 4 park, 5 pull, 6 release = the in-tree library); the share is cumulative.  Book-one and the Cornell box at their sizes, 64 spp.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for sc in "book_one --width 1200 --height 800 --spp 500" "cornell --width 600 --height 600 --spp 1000"; do
 for k in 1 2 3 4 5 6; do
  f=$R/ray-tracer_amd/lib/variants/librt_probe$k.so; [ $k = 6 ] && f=$R/ray-tracer_amd/lib/librt_mi355x.so
  RT_MI355X_LIB=$f timeout -k 10 200 python3 $R/bench.py --scene $sc --steps 1 --warmup 0 --no-cpu-baseline --no-check 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d['roofline']; c=r['block_cycle_share']
print('${sc%% *}', 'up to step $k: swap share of all block cycles', round(c['swap'],4), 'finish', round(c['finish'],3), 'shade', round(c['shade'],3), 'shade executions per sample', round(r['block_executions_per_sample']['shade']/64,4))" || exit 1
 done
done
