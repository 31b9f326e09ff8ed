#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 100 tools/repro/stale_read_two_queues.bin 40 2 > gpurun_out/r4_stale2.out 2>&1; echo "two threads rc $?: $(tail -3 gpurun_out/r4_stale2.out)"
timeout -k 10 100 tools/repro/stale_read_two_queues.bin 15 1 > gpurun_out/r4_stale1.out 2>&1; echo "one thread rc $?: $(tail -2 gpurun_out/r4_stale1.out)"
timeout -k 10 100 tools/repro/stale_read_two_queues.bin 30 4 > gpurun_out/r4_stale4.out 2>&1; echo "four threads rc $?: $(tail -5 gpurun_out/r4_stale4.out)"
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputests3.log 2>&1; rc=$?
tail -4 gpurun_out/r4_gputests3.log
exit 0
