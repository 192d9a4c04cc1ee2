#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats + separate PMC passes (never --pmc together with trace
# domains; the program directly after `--`) of bench.py for one scene.
# Usage: tools/profile_scene.sh <tag> <scene> <W> <H> <spp>     -> gpurun_out/prof_<tag>/summary/
set -u
TAG=$1; SCENE=$2; W=$3; H=$4; SPP=$5
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --scene $SCENE --width $W --height $H --spp $SPP"
echo "python3 bench.py $ARGS" > $OUT/command.txt
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py $ARGS > $OUT/$name.log 2>&1
  echo "$TAG $name rc=$?"
}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/stats.log 2>&1; echo "$TAG stats rc=$?"
run pmc_f64 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE
run pmc_f32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run pmc_wave SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
run pmc_lds SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
run pmc_fetch FETCH_SIZE
run pmc_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 $R/tools/summarize_profile.py $OUT $TAG > $OUT/summary.log 2>&1; echo "$TAG summary rc=$?"
