#!/bin/bash
# GPU box: the three BASELINE scenes at full size with this tree and with every older tree staged under _ab/<name>/ (a copy of
# that commit with its library built in place), interleaved, ROUNDS times.  Same box, same minutes: what a change costs.
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=${1:-2}
for round in $(seq $ROUNDS); do
 for t in $R $R/_ab/*; do
  [ -f "$t/bench.py" ] || continue
  for sc in "book_one --width 1200 --height 800 --spp 500" "cornell --width 600 --height 600 --spp 1000" "cover --width 800 --height 800 --spp 1000"; do
   (cd $t && timeout -k 10 300 python3 bench.py --scene $sc --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null) | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d['roofline']
print('round $round', '$(basename $t)', d['config']['workload'][:12], round(d['value'],1), 'Ms/s kernel_ms', round(r['kernel_ms'],2), 'step_ms', round(d['ms_per_step'],2))"
  done
 done
done
