#!/bin/bash
# the emit form of round 0's last pass: parity suite first, then the headline with and without it (hooks library)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_suffix_sort_gpu.py tests/test_verify_gpu.py tests/test_ref_pins_golden.py -m gpu -x -q > gpurun_out/r4_s19_tests.out 2>&1
rc=$?; tail -5 gpurun_out/r4_s19_tests.out
[ $rc -ne 0 ] && exit $rc
for v in emit noemit; do
  if [ $v = noemit ]; then export KISS_HIP_NO_R0_EMIT=1; fi
  KISS_AMD_LIB=hooks timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-e2e --no-fm --no-exact --no-dm --no-sensitivity > gpurun_out/r4_s19_bench_$v.json 2> gpurun_out/r4_s19_bench_$v.err || { tail -5 gpurun_out/r4_s19_bench_$v.err; exit 1; }
  python3 - $v <<'PY'
import json,sys
j=json.loads(open("gpurun_out/r4_s19_bench_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], j["ms_per_step"], j["roofline"]["kernel_ms_per_step"], j.get("sa_matches_pinned_hash"), j.get("verified"))
PY
done
