#!/bin/bash
# A/B of the swap-at-shade kernel (RT_SWAP=1) against the default one: parity first, then the headline bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export RT_SWAP=1
timeout -k 10 240 python3 -m pytest tests -m gpu -x -q -k "book_one_matches or sharded_render or edge_cases or tiny_images or large_lds or random_scenes or furnace" > gpurun_out/swap_tests.log 2>&1 || { tail -30 gpurun_out/swap_tests.log; exit 1; }
tail -2 gpurun_out/swap_tests.log
for sw in 1 0 1 0; do
  RT_SWAP=$sw timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d['roofline']
print('swap=$sw', round(d['value'],1), 'Ms/s kernel_ms', round(r['kernel_ms'],2), 'util', {k:round(v,2) for k,v in r['simd_utilisation'].items()}, 'exec', {k:round(v,2) for k,v in r['block_executions_per_sample'].items()}, 'cyc', {k:round(v,2) for k,v in r['block_cycle_share'].items()}, 'swap', r.get('swap_at_shade'))" || exit 1
done
