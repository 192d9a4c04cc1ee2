#!/bin/bash
# GPU box: what the per-sample radiance workspace costs in time when it is capped (RT_SAMPLE_WORKSPACE_MB): the headline render in
# 1, 2, 4, 8, 15 passes -- footprint against throughput ("accepted as a time argument, not as a footprint one")
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
 for mb in 0 8192 4096 2048 1024; do
  if [ $mb = 0 ]; then unset RT_SAMPLE_WORKSPACE_MB; else export RT_SAMPLE_WORKSPACE_MB=$mb; fi
  timeout -k 10 300 python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); r=d['roofline']; l=r['launch']
print('round $round cap_MB', '$mb'.rjust(5), 'passes', l['passes'], 'workspace_GB', round(l['workspace_bytes']/1e9,2), 'Msamples/s', round(d['value'],1), 'ms_per_step', round(d['ms_per_step'],2), 'single_render_ms', round(d.get('single_render_ms',0),2), 'match', d.get('image_matches_single_render'), 'oracle', d.get('oracle_pixels'))" || exit 1
 done
done
