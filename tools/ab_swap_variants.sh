#!/bin/bash
# every library under ray-tracer_amd/lib/variants, with and without RT_SWAP (headline bench, swap diagnostics)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; : > $R/gpurun_out/ab_swap.log
if [ -n "$AB_PARITY" ]; then
  for sw in 0 1; do
    RT_SWAP=$sw timeout -k 10 300 python3 -m pytest $R/tests -m gpu -x -q > $R/gpurun_out/swap_tests_$sw.log 2>&1 || { tail -30 $R/gpurun_out/swap_tests_$sw.log; exit 1; }
    tail -1 $R/gpurun_out/swap_tests_$sw.log
  done
fi
for f in $R/ray-tracer_amd/lib/variants/librt_*.so; do
 for sw in ${AB_SWAPS:-0 1}; do
  n=$(basename $f .so)
  RT_SWAP=$sw RT_MI355X_LIB=$f timeout -k 10 120 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; sw=r.get('swap_at_shade') or {}
        print('$n swap=$sw', round(d['value'],1), 'Ms/s kernel_ms', round(r['kernel_ms'],2), 'util', {k:round(v,2) for k,v in r['simd_utilisation'].items()}, 'exec', {k:round(v,2) for k,v in r['block_executions_per_sample'].items()}, 'cyc', {k:round(v,2) for k,v in r['block_cycle_share'].items()}, 'swap', sw)
" >> $R/gpurun_out/ab_swap.log || exit 1
 done
done
cat $R/gpurun_out/ab_swap.log
