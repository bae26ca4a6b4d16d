#!/bin/bash
# GPU box: run a command with a variant build of the library in place.  usage: bash tools/exp_lib.sh build/variants/<lib>.so <command...>
R=$GRAFT_REPO_ROOT; cd $R; L=multi-gpu-path-tracer_amd/libptamd.so; cp $L /tmp/lib_orig.so; cp $1 $L; shift
"$@"; rc=$?
cp /tmp/lib_orig.so $L; exit $rc
