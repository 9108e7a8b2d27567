#!/bin/bash
# Runs GPU steps one after the other on a gpurun box; a step that times out or is killed ends the call (no further GPU
# step after a hang), an ordinary failure does not.  usage: tools/run_steps.sh "<seconds> <command...>" ...
mkdir -p gpurun_out
worst=0
for step in "$@"; do
    secs=${step%% *}
    cmd=${step#* }
    echo "=== [$(date +%T)] (limit ${secs}s) $cmd"
    timeout -k 10 "$secs" bash -c "$cmd"
    rc=$?
    echo "=== rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out / was killed: stopping"; exit $rc; fi
    [ $rc -ne 0 ] && worst=$rc
done
exit $worst
