#!/bin/bash
# SQ counters per kernel (n = 5e8): where do the waves of each kernel spend their cycles?
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/r4_s22_counters.txt 2>&1 || true
grep -o "SQ_[A-Z_0-9]*" gpurun_out/r4_s22_counters.txt | sort -u | tr '\n' ' ' > gpurun_out/r4_s22_sq_names.txt
bash tools/pmc.sh r4_s22_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" 500000000 > gpurun_out/r4_s22_sq1.out 2>&1 || { tail -5 gpurun_out/r4_s22_sq1.out; exit 1; }
bash tools/pmc.sh r4_s22_sq2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" 500000000 > gpurun_out/r4_s22_sq2.out 2>&1 || { tail -5 gpurun_out/r4_s22_sq2.out; exit 1; }
head -30 gpurun_out/r4_s22_sq1.out
