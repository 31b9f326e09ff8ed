set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
B="python bench.py --steps 6 --warmup 3 --no-fm --no-e2e --no-exact --no-dm --cpu-sample 0 --no-fnv"
cp kiss_amd/libkiss_hip.so /tmp/cur.so
for r in 1 2; do
  for v in cur in4 in16; do
    if [ $v = cur ]; then cp /tmp/cur.so kiss_amd/libkiss_hip.so; else cp kiss_amd/libkiss_$v.so.bin kiss_amd/libkiss_hip.so; fi
    $B > gpurun_out/ab7_${v}_$r.json 2> gpurun_out/ab7_${v}_$r.err
  done
done
cp /tmp/cur.so kiss_amd/libkiss_hip.so
echo done
