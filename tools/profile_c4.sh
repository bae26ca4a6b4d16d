#!/bin/bash
# On the GPU box: bash tools/profile_c4.sh <tag> [W H]  -- C4 (8M-triangle terrain, SAH tree) at a reduced
# frame size: kernel stats + HBM traffic passes.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; TAG=$1; W=${2:-480}; H=${3:-270}
ARGS="--config c4 --steps 1 --warmup 0 --no-cpu-baseline --width $W --height $H"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_$TAG.log 2>&1 || echo FAIL stats
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr " " "_" | cut -c1-30)
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$TAG/$N -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_$N.log 2>&1 || echo FAIL $C
done
