#!/bin/bash
# GPU box, round 5: tests/sweeps/random_parity.py over fresh seeds with the round's kernels (cube groups, the binary16 tree in LDS).
#   tools/r05_sweeps.sh <part>     part 1: cover + cubes; part 2: general; part 3: book-one, cameras, scaled, wide, x4
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { # timeout n first depth generator [env...]
  local t=$1 n=$2 first=$3 depth=$4 gen=$5; shift 5
  env "$@" timeout -k 10 $t python3 tests/sweeps/random_parity.py $n $first $depth $gen > gpurun_out/r05_sweep_${gen}_${first}.log 2>&1
  tail -1 gpurun_out/r05_sweep_${gen}_${first}.log | cut -c1-420
}
case "$1" in
1) run 400 2500 9100001 100 cover X=1
   run 400 6000 9200001 40 cubes X=1
   run 400 6000 9300001 40 cubes RT_HALF_NODES=1 ;;
2) run 1100 30000 9400001 40 general X=1 ;;
3) run 300 5000 9500001 100 book_one X=1
   run 300 3000 9600001 100 camera X=1
   run 300 3000 9700001 100 scaled X=1
   run 200 50 9800001 100 wide X=1 ;;
4) run 1100 10000 9900001 100 general X=1 ;;
5) RANDOM_PARITY_SCALE=4 run 900 800 10000001 100 general X=1
   RANDOM_PARITY_SCALE=4 run 200 200 10100001 100 cubes RT_HALF_NODES=1 ;;
# the long versions (fresh seeds again): evidence by weight
6) run 1150 50000 11000001 40 general X=1 ;;
7) run 1150 50000 11100001 40 general X=1 ;;
8) run 600 10000 11200001 100 cover X=1
   run 500 8000 11300001 100 book_one X=1 ;;
9) run 550 15000 11400001 40 cubes X=1
   run 550 15000 11500001 40 cubes RT_HALF_NODES=1 ;;
# more weight where the one explained pixel came from (scaled scenes), cameras, depth 100, four times the size
10) run 600 40000 11600001 100 scaled X=1
    run 450 20000 11700001 100 camera X=1 ;;
11) run 1150 80000 11800001 100 general X=1 ;;
12) RANDOM_PARITY_SCALE=4 run 650 10000 11900001 100 general X=1
    RANDOM_PARITY_SCALE=4 run 450 2000 12000001 100 cubes RT_HALF_NODES=1 ;;
# the families with little weight so far: fields of 41 000 prims (32-bit references), and the cubes generator with a leaf per face
13) run 700 600 12100001 100 wide X=1
    run 300 6000 12200001 40 cubes RT_NO_CUBE_GROUPS=1 ;;
esac
