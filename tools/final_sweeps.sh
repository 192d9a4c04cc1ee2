#!/bin/bash
# GPU box: one more pass of every generator of tests/sweeps/random_parity.py over fresh seeds (the round's last kernels) -> gpurun_out/random_parity_*.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { timeout -k 10 $1 python3 tests/sweeps/random_parity.py $2 $3 100 $4 > gpurun_out/final_sweep_$4.log 2>&1; tail -1 gpurun_out/final_sweep_$4.log | cut -c1-330; }
run 200 1000 8200001 cover
run 200 5000 8300001 book_one
run 200 3000 8400001 camera
run 300 3000 8500001 scaled
run 200 50 8600001 wide
