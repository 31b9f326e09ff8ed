#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r4_glitch_patches.out
for v in vmcnt0 smemwait nop_after_vcmp; do
  echo "== $v" >> gpurun_out/r4_glitch_patches.out
  timeout -k 10 120 tools/repro/glitch_$v.bin 60000 1 0 1 >> gpurun_out/r4_glitch_patches.out 2>&1
  if [ $? -eq 124 ]; then exit 124; fi
done
cat gpurun_out/r4_glitch_patches.out
exit 0
