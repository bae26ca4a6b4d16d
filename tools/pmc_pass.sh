#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/pmc_pass.sh <tag> [bench args...]
# Two separate rocprofv3 --pmc passes over one bench invocation each; CSVs land in gpurun_out/pmc_<tag>/.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
TAG=$1; shift
ARGS="--steps 1 --warmup 0 --no-cpu-baseline $@"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_$TAG/a -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_a.log 2>&1 || echo FAIL a
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_$TAG/b -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_b.log 2>&1 || echo FAIL b
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS SQ_IFETCH SQ_WAVE32_INSTS SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/pmc_$TAG/c -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}_c.log 2>&1 || echo FAIL c
