#!/bin/bash
# on the GPU box: bench.py with the hooks library, alternately without and with ONE environment switch set (A-B of a form the
# hooks build can turn off), two rounds each, same box.  usage: tools/ab_env.sh SWITCH=VALUE [bench args...]
cd ${GRAFT_REPO_ROOT:-.}
sw=$1; shift
for r in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export "$sw"; else unset "${sw%%=*}"; fi
    KISS_AMD_LIB=hooks timeout -k 10 200 python bench.py --steps 10 --no-e2e --no-fm --cpu-sample 0 --no-exact --no-dm --no-sensitivity --profile-steps 2 "$@" > gpurun_out/ab_${v}_$r.json 2> gpurun_out/ab_${v}_$r.err || exit 1
    python -c "import json,sys;j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);k=j['roofline']['kernel_ms_per_step'];print(sys.argv[1], round(j['ms_per_step'],2), {a:round(b,2) for a,b in k.items()}, j.get('sa_matches_pinned_hash'))" gpurun_out/ab_${v}_$r.json
  done
done
