#!/bin/bash
# GPU box: the cover (800x800, 300 spp) with class queues of 128 (what fits), 96, 64, 48 entries, interleaved, 2 rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
 for cap in 128 96 64 48; do
  RT_SWAP_CAP_LIMIT=$cap timeout -k 10 200 python3 $R/tools/cover_lds_probe.py --full --spp 300 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); a=d['default']
print('round $round cap limit $cap', 'ms', round(a['kernel_ms_median'],2), a['kernel_ms'][1:], 'cap', a['swap_cap'], 'lds', a['lds_bytes'], 'lds_nodes', a['lds_nodes'])" || exit 1
 done
done
