#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for mode in 1 0; do
  timeout -k 10 120 tools/repro/vcc_salu_hazard.bin 20000 $mode > gpurun_out/r4_hazard_$mode.out 2>&1; rc=$?
  echo "hazard probe noise=$mode rc $rc: $(cat gpurun_out/r4_hazard_$mode.out)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
KISS_AMD_LIB=hooks timeout -k 10 400 bash tools/prof.sh r4diag --steps 3 --warmup 1 --no-e2e --no-fm --no-dm --no-exact --no-sensitivity --no-fnv --cpu-sample 0 --no-profile > gpurun_out/r4diag_summary.txt 2>&1
head -30 gpurun_out/r4diag_summary.txt
exit 0
