#!/bin/bash
# on the GPU box: bench.py alternately with the library in the tree and with a variant build (kiss_amd/libkiss_old.so.bin),
# two rounds each, same box.  usage: tools/ab_libs.sh [bench args...]
cd ${GRAFT_REPO_ROOT:-.}
cp kiss_amd/libkiss_hip.so /tmp/new.so
for r in 1 2; do
  for v in new old; do
    if [ $v = old ]; then cp kiss_amd/libkiss_old.so.bin kiss_amd/libkiss_hip.so; else cp /tmp/new.so kiss_amd/libkiss_hip.so; fi
    timeout -k 10 200 python bench.py --steps 10 --no-e2e --no-fm --cpu-sample 0 --profile-steps 0 "$@" > gpurun_out/ab_${v}_$r.json 2> gpurun_out/ab_${v}_$r.err || exit 1
  done
done
cp /tmp/new.so kiss_amd/libkiss_hip.so
