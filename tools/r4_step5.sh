#!/bin/bash
# round 4: GPU suite on the current tree; pairs along their diagonals against the walks (hooks build A-B); the two-context
# stress on the hooks build with k_near_tie_runs compiled as shipped and the recorder in the two kernels after it
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputests2.log 2>&1; rc=$?
tail -4 gpurun_out/r4_gputests2.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
B="--steps 3 --warmup 1 --no-e2e --no-fm --no-dm --no-exact --no-sensitivity --no-fnv --cpu-sample 0 --profile-all"
for v in diag walk; do
  if [ $v = walk ]; then export KISS_HIP_NO_PAIR_DIAG=1; else unset KISS_HIP_NO_PAIR_DIAG; fi
  KISS_AMD_LIB=hooks timeout -k 10 200 python bench.py $B > gpurun_out/r4_pairs_$v.json 2> gpurun_out/r4_pairs_$v.err; rc=$?
  echo "pairs $v rc $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_pairs_$v.json").read().strip().splitlines()[-1])
k=d["roofline"]["kernel_ms_per_step"]
print("$v: ms_per_step %.2f segrank %.3f radix_scatter %.3f keygather %.3f flag_compact %.3f lms_sort %.2f verified %s" % (d["ms_per_step"], k["segrank"], k["radix_scatter"], k["keygather"], k["flag_compact"], d["config"]["stage_ms_per_step"]["lms_sort"], d.get("verified")))
PY
done
unset KISS_HIP_NO_PAIR_DIAG
echo "== hooks build, k_near_tie_runs as shipped, recorder in the mark and table kernels" > gpurun_out/r4_trace_noruns.out
KISS_AMD_LIB=default KISS_AMD_LIB_PATH=$PWD/kiss_amd/libkiss_hooks_noruns.so.bin KISS_HIP_NO_SERIALIZE=1 KISS_HIP_TIE_TRACE=1 LX_WARM=1 timeout -k 10 450 python tools/lx_repro.py 4 2 800 >> gpurun_out/r4_trace_noruns.out 2> gpurun_out/r4_trace_noruns.err
echo "rc $?" >> gpurun_out/r4_trace_noruns.out
tail -3 gpurun_out/r4_trace_noruns.out | cut -c1-200
exit 0
