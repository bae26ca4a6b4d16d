#!/bin/bash
# GPU box: C3 (reference tree) time of the stack walk with dealt leaves over {waves/SIMD build under build/variants/libptamd_ds<w>.so} x {PTAMD_DEAL_MIN}.
R=$GRAFT_REPO_ROOT; cd $R
run() { timeout -k 10 300 python3 bench.py --config c3 --walk 5 --steps 2 --warmup 2 --no-cpu-baseline --no-host-visible 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.0f' % d['roofline']['kernel_ms_avg'])"; }
for m in ${MINS:-48}; do echo "waves 5 deal_min $m: $(PTAMD_DEAL_MIN=$m run) ms"; done
for w in ${WAVES:-6 7}; do for m in ${MINS:-48}; do echo "waves $w deal_min $m: $(PTAMD_DEAL_MIN=$m bash tools/exp_lib.sh build/variants/libptamd_ds$w.so bash -c "$(declare -f run); run") ms"; done; done
