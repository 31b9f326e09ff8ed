#!/bin/bash
# on the GPU box: times the radix sort with the normal library, then with the phase-profiling variant
cd ${GRAFT_REPO_ROOT:-.}
N=${1:-100000000}
timeout -k 10 200 python tools/radix_probe.py $N > gpurun_out/rxprobe.log 2>&1 || exit 1
cp kiss_amd/libkiss_hip.so /tmp/n.so && cp kiss_amd/libkiss_prof.so.bin kiss_amd/libkiss_hip.so
timeout -k 10 200 python tools/radix_probe.py $N > gpurun_out/rxprof.log 2>&1
cp /tmp/n.so kiss_amd/libkiss_hip.so
sleep 1
cat gpurun_out/rxprobe.log gpurun_out/rxprof.log
