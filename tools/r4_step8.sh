#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
echo "== asm recorder (re-reads, plain and coherent) behind the shipped k_near_tie_runs" > gpurun_out/r4_asmrec2.out
KISS_AMD_LIB=default KISS_AMD_LIB_PATH=$PWD/kiss_amd/libkiss_hooks_asmrec.so.bin KISS_HIP_NO_SERIALIZE=1 KISS_HIP_TIE_TRACE=1 LX_WARM=1 timeout -k 10 500 python tools/lx_repro.py 4 2 800 >> gpurun_out/r4_asmrec2.out 2> gpurun_out/r4_asmrec2.err
echo "rc $?" >> gpurun_out/r4_asmrec2.out
tail -2 gpurun_out/r4_asmrec2.out | cut -c1-200
grep "unmarked 19\|unmarked 2[0-9]" gpurun_out/r4_asmrec2.err | head -4 | cut -c1-1800
grep "n 400000 k 512" gpurun_out/r4_asmrec2.err | head -1 | cut -c1-1800
timeout -k 10 300 python bench.py --multi-abi 0,0 --steps 3 --warmup 1 --no-e2e --no-fm --no-dm --no-exact --no-sensitivity --no-fnv --cpu-sample 0 > gpurun_out/r4_multi_abi_two_shares.json 2> gpurun_out/r4_multi_abi_two_shares.err
echo "multi-abi rc $?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_multi_abi_two_shares.json").read().strip().splitlines()[-1])
print("multi-abi 0,0: ms_per_step %.2f verified %s phases %s" % (d["ms_per_step"], d.get("verified"), d["config"].get("multi_phase_ms")))
PY
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-e2e --no-fm --no-dm --no-exact --no-sensitivity --no-fnv --cpu-sample 0 > gpurun_out/r4_bench2.json 2> gpurun_out/r4_bench2.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_bench2.json").read().strip().splitlines()[-1])
print("default: ms_per_step %.2f verified %s stage %s" % (d["ms_per_step"], d.get("verified"), d["config"]["stage_ms_per_step"]))
PY
exit 0
