#!/bin/bash
# shapes of the one-pass flag + compaction of round 0 (lms_sort.hip: KISS_FC1_THREADS x KISS_FC1_ITEMS), one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
out=gpurun_out/r4_s20_fc_shapes.log; : > $out
for v in default 512_16_4 256_32_4 256_16_4 512_8_4 default; do
  if [ $v = default ]; then unset KISS_AMD_LIB_PATH; else export KISS_AMD_LIB_PATH=$PWD/kiss_amd/libkiss_fc_$v.so.bin; fi
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-e2e --no-fm --no-exact --no-dm --no-sensitivity > gpurun_out/r4_s20_$v.json 2> gpurun_out/r4_s20_$v.err || { tail -5 gpurun_out/r4_s20_$v.err; exit 1; }
  python3 - $v >> $out <<'PY'
import json,sys
j=json.loads(open("gpurun_out/r4_s20_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
k=j["roofline"]["kernel_ms_per_step"]
print("%-10s ms_per_step %.2f flag_compact %.3f radix_scatter %.2f segrank %.2f hash_ok %s"%(sys.argv[1], j["ms_per_step"], k["flag_compact"], k["radix_scatter"], k["segrank"], j.get("sa_matches_pinned_hash")))
PY
done
cat $out
