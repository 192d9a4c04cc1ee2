#!/bin/bash
# bench the in-tree library under different environment settings (GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; : > $R/gpurun_out/ab_env.log
run() { # label env...
  label=$1; shift
  env "$@" timeout -k 10 120 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$label', round(d['value'],1), 'Ms/s kernel_ms', round(r['kernel_ms'],2), 'util', {k:round(v,3) for k,v in r['simd_utilisation'].items()}, 'cyc', {k:round(v,3) for k,v in r['block_cycle_share'].items()})
" >> $R/gpurun_out/ab_env.log
}
run lds X=1
run nolds RT_NO_LDS_NODES=1
run lds X=1
run nolds RT_NO_LDS_NODES=1
cat $R/gpurun_out/ab_env.log
