#!/bin/bash
# on the GPU box: times the radix sort (tools/radix_probe.py) with the normal library and then with every variant
# kiss_amd/libkiss_*.so.bin (e.g. the phase-profiling build of tools/build_prof.sh), all on the same GPU; variants are
# loaded by path (KISS_AMD_LIB_PATH), the shipped library file is never overwritten
cd ${GRAFT_REPO_ROOT:-.}
N=${1:-100000000}
echo "== normal" > gpurun_out/rx_ab.log
timeout -k 10 200 python tools/radix_probe.py $N >> gpurun_out/rx_ab.log 2>&1 || exit 1
for v in kiss_amd/libkiss_*.so.bin; do
  [ -f "$v" ] || continue
  echo "== $v" >> gpurun_out/rx_ab.log
  KISS_AMD_LIB_PATH=$PWD/$v timeout -k 10 200 python tools/radix_probe.py $N >> gpurun_out/rx_ab.log 2>&1
done
echo "== normal again" >> gpurun_out/rx_ab.log
timeout -k 10 200 python tools/radix_probe.py $N >> gpurun_out/rx_ab.log 2>&1
sleep 1
cat gpurun_out/rx_ab.log
