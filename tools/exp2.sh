#!/bin/bash
# GPU box: dealt walk = default for the box.  Whole GPU suite, C2 profile, tail measurement, shares.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q > gpurun_out/exp2_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/exp2_tests.log
bash tools/profile_c2.sh r02b_c2 --config c2 --no-host-visible; echo "profile c2 done"
cd $R
run() { timeout -k 10 300 python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-host-visible "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f ms  %s' % (d['roofline']['kernel_ms_avg'], d['roofline']['kernel']))"; }
for h in 540 1080 2160 4320; do echo "c2 height $h: $(run --config c2 --height $h)"; done | tee gpurun_out/exp2_tail.txt
for h in 1080 2160; do echo "c2 mask walk height $h: $(run --config c2 --height $h --walk 2)"; done | tee -a gpurun_out/exp2_tail.txt
echo "c5: $(run --config c5)" | tee -a gpurun_out/exp2_tail.txt
echo "c1: $(run --config c1)" | tee -a gpurun_out/exp2_tail.txt
for n in 2 4 8; do timeout -k 10 300 python3 bench.py --config c2 --share-of $n --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1; done | tee gpurun_out/exp2_shares.txt
