#!/bin/bash
# run bench.py for every library variant in ray-tracer_amd/lib/variants (GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
: > $R/gpurun_out/ab.log
for rep in 1; do
for f in $R/ray-tracer_amd/lib/variants/librt_*.so; do
  n=$(basename $f .so)
  RT_MI355X_LIB=$f timeout -k 10 120 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$n rep$rep', round(d['value'],1), 'Ms/s kernel_ms', round(r['kernel_ms'],2), 'util', {k:round(v,3) for k,v in r['simd_utilisation'].items()}, 'exec/sample', {k:round(v,2) for k,v in r['block_executions_per_sample'].items()}, 'cyc', {k:round(v,3) for k,v in r['block_cycle_share'].items()})
" >> $R/gpurun_out/ab.log
done
done
cat $R/gpurun_out/ab.log
