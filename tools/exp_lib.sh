#!/bin/bash
# GPU box: run a command with a variant build of the library.  usage: bash tools/exp_lib.sh build/variants/<lib>.so <command...>
# The variant is loaded through PTAMD_LIBRARY (multi-gpu-path-tracer_amd/_capi.py); the product library is never touched.
R=$GRAFT_REPO_ROOT; cd $R; V=$(readlink -f $1); shift
PTAMD_LIBRARY=$V "$@"
