#!/bin/bash
# GPU box: tools/tile_order_probe.py with the test-hooks library over the experiment's knobs (cost levels; a whole image too)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export RT_MI355X_LIB=$R/ray-tracer_amd/lib/librt_mi355x_testhooks.so RT_TEST_TILE_ORDER_WHOLE=1
for L in 0 4 8 16 64; do
  echo "== levels $L"
  RT_TEST_TILE_ORDER_LEVELS=$L timeout -k 10 300 python3 $R/tools/tile_order_probe.py --worlds 1,8 --repeats 5 --out $R/gpurun_out/r04_tile_order_levels_$L.json \
    | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['scene'], r['shard'], 'asc %.2f learnt %.2f gain %.2f%%' % (r['ascending_kernel_ms'], r['learnt_kernel_ms'], 100 * r['kernel_gain']))
" || exit 1
done
