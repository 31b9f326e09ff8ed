#!/bin/bash
# round 4: (1) round-0 flag + compaction forms (KISS_HIP_FC0_FORM, hooks build) on the headline text; (2) DESIGN 4.2: the
# two-context stress on three variants of the shipped build without the device lock, each with ONE of the three near-end
# tie kernels compiled as in the hooks build (which never showed the fault)
set -o pipefail
mkdir -p gpurun_out
for f in 1 2 3 4 5; do
  KISS_AMD_LIB=hooks KISS_HIP_FC0_FORM=$f timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-e2e --no-fm --no-dm --no-exact --no-sensitivity --no-fnv --cpu-sample 0 --profile-all > gpurun_out/r4_fc0_form$f.json 2> gpurun_out/r4_fc0_form$f.err
  rc=$?; echo "fc0 form $f rc $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4_fc0_form$f.json").read().strip().splitlines()[-1])
print("form $f: ms_per_step %.2f flag_compact %.3f verified %s" % (d["ms_per_step"], d["roofline"]["kernel_ms_per_step"]["flag_compact"], d.get("verified")))
PY
done
run() { # name, lib path
  echo "== $1" > gpurun_out/r4_abk_$1.out; date >> gpurun_out/r4_abk_$1.out
  KISS_AMD_LIB=default KISS_AMD_LIB_PATH=$2 LX_WARM=1 timeout -k 10 450 python tools/lx_repro.py 4 2 800 >> gpurun_out/r4_abk_$1.out 2> gpurun_out/r4_abk_$1.err
  rc=$?; echo "rc $rc" >> gpurun_out/r4_abk_$1.out; date >> gpurun_out/r4_abk_$1.out
  tail -3 gpurun_out/r4_abk_$1.out | cut -c1-200
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
}
run MARK $PWD/kiss_amd/libkiss_nolock_MARK.so.bin
run RUNS $PWD/kiss_amd/libkiss_nolock_RUNS.so.bin
run TABLE $PWD/kiss_amd/libkiss_nolock_TABLE.so.bin
exit 0
