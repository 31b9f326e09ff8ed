# final validation of the round: whole GPU suite, the driver's bench command, a short differential fuzz on the final library
set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/t_final.log 2>&1; echo "rc=$?" >> gpurun_out/t_final.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_final.json 2> gpurun_out/r3_final.err; echo "bench rc=$?" >> gpurun_out/t_final.log
python tools/fuzz_parity.py 150 301 > gpurun_out/fuzz_final.log 2>&1; echo "fuzz rc=$?" >> gpurun_out/t_final.log
python tools/fuzz_sharded.py 8 302 >> gpurun_out/fuzz_final.log 2>&1; echo "fuzz_sharded rc=$?" >> gpurun_out/t_final.log
tail -n 6 gpurun_out/t_final.log; tail -n 3 gpurun_out/fuzz_final.log
