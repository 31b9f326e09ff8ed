set -o pipefail
python tools/fuzz_fm.py 100 303 > gpurun_out/fuzz_fm_final.log 2>&1; echo "fuzz_fm rc=$?"
python tools/fuzz_cli.py 12 304 > gpurun_out/fuzz_cli_final.log 2>&1; echo "fuzz_cli rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_final2.json 2> gpurun_out/r3_final2.err; echo "bench rc=$?"
bash tools/prof.sh r03b --steps 3 --warmup 1 --no-e2e --no-fm --no-exact --no-dm --cpu-sample 0 --profile-steps 0 --no-verify --no-fnv > gpurun_out/r03b_summary.txt 2>&1
tail -n 2 gpurun_out/fuzz_fm_final.log gpurun_out/fuzz_cli_final.log
