#!/bin/bash
# (CPU, after tools/r05_final.sh a / b / c) copy what is cited from gpurun_out/ into profiles/
cd "$(dirname "$0")/.."
case "$1" in
a) for s in book_one cornell cover; do mkdir -p profiles/r05_$s; cp gpurun_out/prof_r05_$s/summary/summary.json gpurun_out/prof_r05_$s/summary/kernel_stats.csv profiles/r05_$s/; done
   cp gpurun_out/r05_gputest_final.log profiles/r05_logs/gputest_final.log ;;
b) for s in book_one cornell cover; do cp gpurun_out/r05_bench_$s.log profiles/r05_bench_$s.log; done
   cp gpurun_out/configs.json profiles/r05_configs.json
   cp gpurun_out/r05_shard_scaling.log gpurun_out/r05_shard_scaling_ascending.log profiles/
   cp gpurun_out/random_parity_cover_from_9100001_depth_100.json profiles/r05_random_parity_cover_2500.json
   cp gpurun_out/random_parity_cubes_from_9200001_depth_40.json profiles/r05_random_parity_cubes_6000.json
   cp gpurun_out/random_parity_cubes_from_9300001_depth_40.json profiles/r05_random_parity_cubes_6000_binary16_tree.json
   cp gpurun_out/random_parity_book_one_from_9500001_depth_100.json profiles/r05_random_parity_book_one_5000.json
   cp gpurun_out/random_parity_camera_from_9600001_depth_100.json profiles/r05_random_parity_camera_3000.json
   cp gpurun_out/random_parity_scaled_from_9700001_depth_100.json profiles/r05_random_parity_scaled_3000.json
   cp gpurun_out/random_parity_wide_from_9800001_depth_100.json profiles/r05_random_parity_wide_50.json ;;
c) cp gpurun_out/random_parity_general_from_9400001_depth_40.json profiles/r05_random_parity_general_30k.json
   cp gpurun_out/random_parity_general_from_10000001_depth_100_x4.json profiles/r05_random_parity_general_x4_800.json
   cp gpurun_out/random_parity_cubes_from_10100001_depth_100_x4.json profiles/r05_random_parity_cubes_x4_200_binary16_tree.json
   cp gpurun_out/random_parity_general_from_9900001_depth_100.json profiles/r05_random_parity_general_10k_depth100.json ;;
esac
