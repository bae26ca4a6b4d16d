#!/bin/bash
# GPU box, repo root.  Round-2 baseline pass: full -m gpu suite, HEAD profiles of C2 and C4, share sweeps, default bench.
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q -s --durations=15 > gpurun_out/r02_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r02_tests.log
tail -5 gpurun_out/r02_tests.log
bash tools/profile_c2.sh r02a_c2 --config c2 --no-host-visible; echo "profile c2 done"
bash tools/profile_c2.sh r02a_c4 --config c4 --no-host-visible; echo "profile c4 done"
cd $R && bash tools/shares_json.sh gpurun_out/r02a_shares.jsonl
cd $R && python3 bench.py > gpurun_out/r02a_bench.log 2>&1; tail -1 gpurun_out/r02a_bench.log | cut -c1-1500
