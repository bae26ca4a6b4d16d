#!/bin/bash
# GPU box: run a command repeatedly; if a run is still alive after LIMIT seconds, say what every thread is blocked in
# (/proc/<pid>/task/*/wchan, syscall) and end that run by its PID.  usage: bash tools/hang_hunt.sh <runs> <limit_s> <command...>
N=$1; LIMIT=$2; shift 2
hangs=0
for i in $(seq 1 $N); do
  "$@" > /tmp/hh_out.txt 2>&1 &
  pid=$!
  t=0
  while kill -0 $pid 2>/dev/null && [ $t -lt $((LIMIT * 10)) ]; do sleep 0.1; t=$((t + 1)); done
  if kill -0 $pid 2>/dev/null; then
    hangs=$((hangs + 1))
    echo "run $i: still alive after $LIMIT s (pid $pid); threads:"
    for task in /proc/$pid/task/*; do
      echo "  tid $(basename $task) comm $(cat $task/comm 2>/dev/null) wchan $(cat $task/wchan 2>/dev/null) syscall $(cut -d' ' -f1 $task/syscall 2>/dev/null) state $(grep State $task/status 2>/dev/null | cut -f2)"
    done
    tail -5 /tmp/hh_out.txt
    kill $pid; sleep 1; kill -9 $pid 2>/dev/null
    break
  fi
  wait $pid; rc=$?
  if [ $rc -ne 0 ]; then echo "run $i: rc $rc"; tail -3 /tmp/hh_out.txt; fi
done
echo "hang hunt: $hangs hang(s) in $i run(s)"
