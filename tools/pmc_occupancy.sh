#!/bin/bash
# On the GPU box: bash tools/pmc_occupancy.sh <tag> [bench args]: wave-occupancy counters per launch
# (first launch = raster order, second = cost order).
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; TAG=$1; shift
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_LEVEL_WAVES --output-format csv -d $R/gpurun_out/occ_$TAG -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/occ_$TAG.log 2>&1 || echo FAIL
