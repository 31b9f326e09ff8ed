#!/bin/bash
# phase profile of k_fc0_onepass (-DFC_PROF variant), headline text
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
KISS_AMD_LIB_PATH=$PWD/kiss_amd/libkiss_fcprof.so.bin timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-e2e --no-fm --no-exact --no-dm --no-sensitivity --no-profile > gpurun_out/r4_s26.json 2> gpurun_out/r4_s26.err || { tail -5 gpurun_out/r4_s26.err; exit 1; }
grep fc_prof gpurun_out/r4_s26.err | tail -3
