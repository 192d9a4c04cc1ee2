#!/bin/bash
# GPU box: book-one 1200x800x500 with the in-tree library and every variant under ray-tracer_amd/lib/variants, interleaved, three rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
for f in $R/ray-tracer_amd/lib/librt_mi355x.so $R/ray-tracer_amd/lib/variants/librt_*.so; do
  [ -f "$f" ] || continue
  RT_MI355X_LIB=$f timeout -k 10 300 python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --check 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d['roofline']
print('$(basename $f .so)', round(d['value'],1), 'Ms/s kernel_ms', round(r['kernel_ms'],2), 'image ok', d.get('image_matches_single_gpu'))"
done
done
