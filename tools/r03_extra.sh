#!/bin/bash
# GPU box, repo root: what the closing pass adds on top of r03_final.sh <tag> a -- C2 profiled ONE frame at a time, the default
# bench line at the driver's step count, whole frames of the other configs at 1 / 2 frames in flight.
T=${1:-r03q}
cd $GRAFT_REPO_ROOT
bash tools/profile_c2.sh ${T}1_c2 --config c2 --no-host-visible --frames-in-flight 1; echo "profile c2 F=1 done"
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench20.log 2>&1
python3 bench.py > gpurun_out/${T}_bench.log 2>&1
for cfg in "c4" "c3 --bvh sah" "c5" "c1"; do for f in 1 2; do python3 bench.py --config $cfg --frames-in-flight $f --steps 6 --warmup 3 --no-cpu-baseline --no-host-visible 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$cfg', d['frames_in_flight'], round(d['ms_per_step'],3), d['value'])"; done; done > gpurun_out/${T}_whole.log 2>&1
cat gpurun_out/${T}_whole.log; tail -1 gpurun_out/${T}_bench20.log | cut -c1-250
