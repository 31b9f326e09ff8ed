#!/bin/bash
# Runs the given commands (one per argument, each a shell string) one after the other on the GPU box, each under its own
# timeout; a step that is killed by its timeout (124/137) ends the whole call -- nothing else is started on the GPU
# after a hang.  A step that merely fails (tests red) does not stop the next one.  Logs go to gpurun_out/.
# usage: tools/gpu_steps.sh SECONDS 'cmd1' 'cmd2' ...
limit=$1; shift
mkdir -p gpurun_out
worst=0
i=0
for cmd in "$@"; do
  i=$((i+1))
  echo "[step $i] $cmd" | tee -a gpurun_out/steps.log
  timeout -k 10 "$limit" bash -c "$cmd"
  rc=$?
  echo "[step $i] exit $rc" | tee -a gpurun_out/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[step $i] killed at its limit: stopping" | tee -a gpurun_out/steps.log; exit $rc; fi
  [ $rc -ne 0 ] && worst=$rc
done
exit $worst
