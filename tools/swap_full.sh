#!/bin/bash
# full GPU parity suite in each swap mode, then every BASELINE config with and without swap in the general kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for mode in "RT_SWAP=1 RT_SWAP_GENERAL=0" "RT_SWAP=1 RT_SWAP_GENERAL=1" "RT_SWAP=0 RT_SWAP_GENERAL=0"; do
  tag=$(echo $mode | tr -d ' =A-Z_')
  env $mode timeout -k 10 300 python3 -m pytest tests -m gpu -x -q > gpurun_out/swap_full_$tag.log 2>&1 || { echo "FAILED in mode $mode"; tail -30 gpurun_out/swap_full_$tag.log; exit 1; }
  echo "$mode: $(tail -1 gpurun_out/swap_full_$tag.log)"
done
for g in 0 1; do
  RT_SWAP_GENERAL=$g timeout -k 10 300 python3 tools/run_configs.py > gpurun_out/configs_g$g.log 2>&1 || { tail -20 gpurun_out/configs_g$g.log; exit 1; }
  cp gpurun_out/configs.json gpurun_out/configs_g$g.json
  python3 -c "
import json
for r in json.load(open('gpurun_out/configs_g$g.json')): print('general_swap=$g', r['config'][:40], round(r['Msamples_per_s'],1), 'Ms/s kernel_ms', round(r['render_kernel_ms'],2), 'mean', r['mean'])"
done
