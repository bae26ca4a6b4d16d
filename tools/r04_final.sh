#!/bin/bash
# GPU box, repo root.  Round-4 closing pass on the frozen kernels, in calls of < 20 minutes each:
#   bash tools/r04_final.sh <tag> a   full -m gpu suite; profiles (kernel trace + PMC passes) of C2 (the default line: two frames in flight,
#                                     frames delivered to the host) and of C2 one frame at a time
#   bash tools/r04_final.sh <tag> b   profiles of C4, C3 through the SAH tree, C3 with the reference tree
#   bash tools/r04_final.sh <tag> c   the sweep between the configs, the default bench line (driver's form and 20 steps), whole frames of every config
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; T=${1:-r04f}; PART=${2:-a}
if [ "$PART" = a ]; then
  python3 -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/${T}_tests.log
  tail -3 gpurun_out/${T}_tests.log
  PROF_STEPS=8 bash tools/profile_c2.sh ${T}_c2 --config c2 --no-host-visible; echo "profile c2 done"
  cd $R && bash tools/profile_c2.sh ${T}1_c2 --config c2 --no-host-visible --frames-in-flight 1; echo "profile c2 F=1 done"
elif [ "$PART" = b ]; then
  PROF_STEPS=8 bash tools/profile_c2.sh ${T}_c4 --config c4 --no-host-visible; echo "profile c4 done"
  cd $R && PROF_STEPS=8 bash tools/profile_c2.sh ${T}_c3sah --config c3 --bvh sah --no-host-visible; echo "profile c3 sah done"
  cd $R && bash tools/profile_c2.sh ${T}_c3ref --config c3 --no-host-visible; echo "profile c3 (reference tree) done"
else
  python3 tools/scene_sweep.py > gpurun_out/${T}_scene_sweep.json 2> gpurun_out/${T}_scene_sweep.log; echo "sweep rc=$?"
  python3 bench.py > gpurun_out/${T}_bench.log 2>&1; tail -1 gpurun_out/${T}_bench.log | cut -c1-200
  python3 bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench20.log 2>&1
  python3 bench.py --host native --steps 10 --warmup 3 > gpurun_out/${T}_bench_native.log 2>&1
  for cfg in "c4" "c3 --bvh sah" "c5" "c1"; do for f in 1 2; do python3 bench.py --config $cfg --frames-in-flight $f --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$cfg', d['frames_in_flight'], round(d['ms_per_step'],3), d.get('ms_per_frame_latency'), d['value'])"; done; done > gpurun_out/${T}_whole.log 2>&1
  cat gpurun_out/${T}_whole.log
fi
