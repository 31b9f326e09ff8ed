#!/bin/bash
# on the GPU box: bench.py alternately with the library in the tree and with a variant build named by path
# (KISS_AMD_LIB_PATH, kiss_amd/_lib.py; default kiss_amd/libkiss_old.so.bin), two rounds each, same box.  The shipped
# library file is never overwritten.  usage: [VARIANT=path] tools/ab_libs.sh [bench args...]
cd ${GRAFT_REPO_ROOT:-.}
VARIANT=${VARIANT:-$PWD/kiss_amd/libkiss_old.so.bin}
for r in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export KISS_AMD_LIB_PATH=$VARIANT; else unset KISS_AMD_LIB_PATH; fi
    timeout -k 10 200 python bench.py --steps 10 --no-e2e --no-fm --cpu-sample 0 --profile-steps 0 "$@" > gpurun_out/ab_${v}_$r.json 2> gpurun_out/ab_${v}_$r.err || exit 1
  done
done
