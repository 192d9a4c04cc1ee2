#!/bin/bash
# GPU box: render_kernel ms of the given scene(s) with the in-tree library and the named variants, interleaved, 2 rounds.
#   tools/ab_three_scenes.sh "<scene args>" <variant names...>      e.g.  "book_one --width 1200 --height 800 --spp 500" book1024
R=${GRAFT_REPO_ROOT:-$(pwd)}
sc=$1; shift
for round in 1 2; do
 for v in mi355x "$@"; do
  f=$R/ray-tracer_amd/lib/variants/librt_$v.so; [ $v = mi355x ] && f=$R/ray-tracer_amd/lib/librt_mi355x.so
  RT_MI355X_LIB=$f timeout -k 10 300 python3 $R/bench.py --scene $sc --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d['roofline']
print('round $round', '$v'.ljust(14), d['config']['workload'][:12], round(d['value'],1), 'Ms/s step', round(d['ms_per_step'],2), 'kernel_ms', round(r['kernel_ms'],2), 'block', r['launch']['block_threads'], 'lds', r['launch']['lds_bytes'], 'cap', r['launch']['swap_cap'], 'match', d.get('image_matches_single_render'))" || exit 1
 done
done
