# round-3 evidence runs (one gpurun call): kernel trace, PMC traffic, sharded / multi-device phase timings, CLI end to end
set -o pipefail
bash tools/prof.sh r03 --steps 3 --warmup 1 --no-e2e --no-fm --no-exact --no-dm --cpu-sample 0 --profile-steps 0 --no-verify --no-fnv > gpurun_out/r03_summary.txt 2>&1
bash tools/pmc.sh r03_fetch "FETCH_SIZE" 500000000 genome > gpurun_out/r03_fetch.txt 2>&1
bash tools/pmc.sh r03_write "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" 500000000 genome > gpurun_out/r03_write.txt 2>&1
python bench.py --steps 3 --warmup 1 --force-sharded --sharded-timings --no-fnv --cpu-sample 0 > gpurun_out/r03_sharded_one_rank.json 2> gpurun_out/r03_sharded_one_rank.err
python bench.py --steps 3 --warmup 1 --multi-abi 0,0 --no-fm --no-e2e --no-exact --no-dm --cpu-sample 0 --no-fnv > gpurun_out/r03_multi_abi_two_shares.json 2> gpurun_out/r03_multi_abi_two_shares.err
python bench.py --steps 3 --warmup 1 --multi-abi 0 --no-fm --no-e2e --no-exact --no-dm --cpu-sample 0 --no-fnv > gpurun_out/r03_multi_abi_one_device.json 2> gpurun_out/r03_multi_abi_one_device.err
python tools/cli_e2e.py 3117292070 --devices 0,0 > gpurun_out/r03_cli_e2e_two_shares.log 2>&1
head -n 30 gpurun_out/r03_summary.txt
