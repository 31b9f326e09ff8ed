#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/prof.sh r4_s24 --steps 4 --warmup 1 --cpu-sample 0 --no-e2e --no-fm --no-exact --no-dm --no-sensitivity --no-profile > gpurun_out/r4_s24_summary.txt 2>&1 || { tail gpurun_out/r4_s24_summary.txt; exit 1; }
head -45 gpurun_out/r4_s24_summary.txt
