#!/bin/bash
# builds kiss_amd/libkiss_hip.so and phase-profiling variants: of the radix scatter (kiss_amd/libkiss_prof.so.bin,
# -DRX_PROF: per-phase wall-clock ticks of thread 0 of every tile, printed by kiss_radix_check)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/kiss_amd/csrc
make 2>&1 | grep -E "error|warning" || true
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -DRX_PROF $EXTRA -c radix.hip -o /tmp/radix_prof.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libkiss_prof.so.bin api.o scan.o classify.o /tmp/radix_prof.o lms_sort.o isa.o place.o induce.o fm.o stages.o multi.o fasta.o general.o verify.o xfer.o
# the same for the one-pass flag + compaction of round 0 (-DFC_PROF: lms_sort.hip prints "[fc_prof] ..." after the pass)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -DFC_PROF $EXTRA -c lms_sort.hip -o /tmp/lms_fcprof.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libkiss_fcprof.so.bin api.o scan.o classify.o radix.o /tmp/lms_fcprof.o isa.o place.o induce.o fm.o stages.o multi.o fasta.o general.o verify.o xfer.o
ls -la ../libkiss_prof.so.bin ../libkiss_fcprof.so.bin ../libkiss_hip.so
