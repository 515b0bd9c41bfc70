#!/bin/bash
# One gpurun call = a list of steps run one after the other, each with its own timeout and log under gpurun_out/ (the tail of each is echoed):
#   tools/gpu_call.sh "name1:timeout1:command1" "name2:timeout2:command2" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for step in "$@"; do
    name="${step%%:*}"; rest="${step#*:}"; to="${rest%%:*}"; cmd="${rest#*:}"
    echo "=== $name (timeout $to s): $cmd"
    ( time timeout "$to" bash -c "$cmd" ) > "gpurun_out/$name.log" 2>&1
    echo "rc=$? ($name)"
    tail -n 12 "gpurun_out/$name.log" | cut -c1-1800
done
