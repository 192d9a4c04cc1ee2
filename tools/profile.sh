#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats + separate PMC passes of bench.py
# (never --pmc together with trace domains).  Usage: tools/profile.sh <tag> [bench args]
set -u
TAG=${1:-run}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B "$@" > $OUT/stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -- $B "$@" > $OUT/pmc1.log 2>&1; echo "pmc1 rc=$?"
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- $B "$@" > $OUT/pmc2.log 2>&1; echo "pmc2 rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- $B "$@" > $OUT/pmc3.log 2>&1; echo "pmc3 rc=$?"
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -- $B "$@" > $OUT/pmc4.log 2>&1; echo "pmc4 rc=$?"
python3 $R/tools/summarize_profile.py $OUT $TAG
