#!/bin/bash
# GPU box: the cover (800x800, 300 spp) with the round-5 structure and with each piece switched off (interleaved, 2 rounds)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
 for env in "" "RT_NO_HALF_NODES=1" "RT_NO_CUBE_GROUPS=1" "RT_NO_CUBE_GROUPS=1 RT_NO_HALF_NODES=1"; do
  echo "== round $round [$env]"
  env $env timeout -k 10 300 python3 $R/tools/cover_lds_probe.py --full --spp 300 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); a=d['default']
print('nodes', d['n_nodes'], 'ms', round(a['kernel_ms_median'],2), 'lds_nodes', a['lds_nodes'], 'block', a['block_threads'], 'lds_bytes', a['lds_bytes'], 'cap', a['swap_cap'], '| no-lds ms', round(d['RT_NO_LDS_NODES=1']['kernel_ms_median'],2), 'same image', d['RT_NO_LDS_NODES=1']['same_image_as_first_mode'])" || exit 1
 done
done
