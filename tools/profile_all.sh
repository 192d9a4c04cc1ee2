#!/bin/bash
# GPU box: the rocprofv3 passes of tools/profile_scene.sh for the three BASELINE scenes at their full sizes
# -> gpurun_out/prof_{book_one,cornell,cover}/summary/{summary.json,kernel_stats.csv}; copy those into profiles/<round>_<scene>/ (ROUND=r05 by default)
R=${GRAFT_REPO_ROOT:-$(pwd)}
$R/tools/profile_scene.sh ${ROUND:-r05}_book_one book_one 1200 800 500 && \
$R/tools/profile_scene.sh ${ROUND:-r05}_cornell cornell 600 600 1000 && \
$R/tools/profile_scene.sh ${ROUND:-r05}_cover cover 800 800 1000
