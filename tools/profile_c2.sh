#!/bin/bash
# On the GPU box, from the repo root: bash tools/profile_c2.sh <tag> [bench args, e.g. --config c3 --bvh sah]
# kernel-trace/stats run of the bench (default: C2) + separate PMC passes (HBM traffic, SQ counters).
# Afterwards, here: python tools/make_profile_summary.py <tag> <cfg>  ->  profiles/<tag>_<cfg>_*.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; TAG=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps ${PROF_STEPS:-3} --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$TAG.log 2>&1 || echo FAIL stats
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$TAG/$C -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_${TAG}_$C.log 2>&1 || echo FAIL $C
done
cd $R && bash tools/pmc_pass.sh $TAG "$@"
