#!/bin/bash
# GPU box, repo root.  Round-2 closing pass on the frozen kernels: full -m gpu suite, profiles of C2 / C3-SAH / C3 with the reference tree / C4 (kernel
# trace + PMC passes), share sweeps, wave timeline, default bench line.  usage: bash tools/r02_final.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; T=${1:-r02f}
python3 -m pytest tests -m gpu -x -q -s --durations=8 > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/${T}_tests.log
tail -3 gpurun_out/${T}_tests.log
bash tools/profile_c2.sh ${T}_c2 --config c2 --no-host-visible; echo "profile c2 done"
cd $R && bash tools/profile_c2.sh ${T}_c4 --config c4 --no-host-visible; echo "profile c4 done"
cd $R && bash tools/profile_c2.sh ${T}_c3sah --config c3 --bvh sah --no-host-visible; echo "profile c3 sah done"
cd $R && bash tools/profile_c2.sh ${T}_c3ref --config c3 --no-host-visible; echo "profile c3 (reference tree) done"
cd $R && bash tools/shares_json.sh gpurun_out/${T}_shares.jsonl
cd $R && bash tools/timeline.sh ${T} c2
cd $R && bash tools/lone_tile.sh c4 3 > gpurun_out/${T}_c4_lone_tiles.txt 2>&1; bash tools/timeline.sh ${T}_c4 c4
cd $R && python3 bench.py > gpurun_out/${T}_bench.log 2>&1; tail -1 gpurun_out/${T}_bench.log
