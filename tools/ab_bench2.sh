#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; : > $R/gpurun_out/ab.log
for f in $R/ray-tracer_amd/lib/variants/librt_*.so; do
 for e in X=1 RT_NO_LDS_NODES=1; do
  n=$(basename $f .so)
  env $e RT_MI355X_LIB=$f timeout -k 10 120 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$n $e', round(d['value'],1), 'Ms/s kernel_ms', round(r['kernel_ms'],2), 'util', {k:round(v,3) for k,v in r['simd_utilisation'].items()}, 'cyc', {k:round(v,3) for k,v in r['block_cycle_share'].items()})
" >> $R/gpurun_out/ab.log
 done
done
cat $R/gpurun_out/ab.log
