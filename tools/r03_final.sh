#!/bin/bash
# GPU box, repo root.  Round-3 closing pass on the frozen kernels, in two calls (a gpurun call is capped at 20 minutes):
#   bash tools/r03_final.sh <tag> a   full -m gpu suite, profiles (kernel trace + PMC passes) of C2 / C4 / C3-SAH / C3 with the reference tree
#   bash tools/r03_final.sh <tag> b   share sweeps, wave timelines of C2 and C4, lane census of the SAH kernel, default bench line
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; T=${1:-r03f}; PART=${2:-a}
if [ "$PART" = a ]; then
  python3 -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/${T}_tests.log
  tail -3 gpurun_out/${T}_tests.log
  bash tools/profile_c2.sh ${T}_c2 --config c2 --no-host-visible; echo "profile c2 done"
  cd $R && bash tools/profile_c2.sh ${T}_c4 --config c4 --no-host-visible; echo "profile c4 done"
  cd $R && bash tools/profile_c2.sh ${T}_c3sah --config c3 --bvh sah --no-host-visible; echo "profile c3 sah done"
  cd $R && bash tools/profile_c2.sh ${T}_c3ref --config c3 --no-host-visible; echo "profile c3 (reference tree) done"
else
  bash tools/shares_json.sh gpurun_out/${T}_shares.jsonl
  cd $R && bash tools/timeline.sh ${T} c2
  cd $R && bash tools/timeline.sh ${T}_c4 c4
  cd $R && bash tools/occupancy_census.sh c4 c3 > gpurun_out/${T}_census.log 2>&1; cp gpurun_out/occupancy_census.json gpurun_out/${T}_census.json
  cd $R && python3 bench.py > gpurun_out/${T}_bench.log 2>&1; tail -1 gpurun_out/${T}_bench.log
fi
