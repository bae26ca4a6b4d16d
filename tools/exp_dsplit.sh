#!/bin/bash
# GPU box: kernel time of one rank's share of an N-rank frame against the split of dear tiles (env VAR, default PTAMD_DEAL_SPLIT).
# usage: [VAR=PTAMD_SAH_SPLIT] bash tools/exp_dsplit.sh "<parts>" "<share counts>" [bench args]
R=$GRAFT_REPO_ROOT; cd $R; VAR=${VAR:-PTAMD_DEAL_SPLIT}
for h in $1; do
  for n in $2; do
    if [ $n = 1 ]; then
      r=$(env $VAR=$h timeout -k 10 300 python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-host-visible ${@:3} 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f' % d['roofline']['kernel_ms_avg'])")
    else
      r=$(env $VAR=$h timeout -k 10 300 python3 bench.py --share-of $n --steps 4 --warmup 3 --no-cpu-baseline ${@:3} 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f' % d['kernel_ms'])")
    fi
    echo "$VAR=$h share 1/$n: $r ms"
  done
done
