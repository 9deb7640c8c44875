set -e
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_radix_tier.py tests/test_gpu_q1_large.py -m gpu -x -q > gpurun_out/r04/gputest_subset.log 2>&1 || { tail -60 gpurun_out/r04/gputest_subset.log; exit 1; }
tail -2 gpurun_out/r04/gputest_subset.log
( timeout -k 10 300 python tools/bench_radix.py 67108864 4194304 1 5; RADIX_BENCH_COUNT=1 timeout -k 10 300 python tools/bench_radix.py 67108864 4194304 1 4; RADIX_BENCH_STR=12 timeout -k 10 300 python tools/bench_radix.py 67108864 4194304 1 4; timeout -k 10 300 python tools/bench_radix.py 600037902 500000 287 3; timeout -k 10 300 python tools/bench_radix.py 67108864 16777216 1 4 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/radix_tier_64M.txt
cat gpurun_out/r04/radix_tier_64M.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/rxstats -- python3 tools/bench_radix.py 67108864 4194304 1 5 > /dev/null 2>&1
cp $(find gpurun_out/r04/rxstats -name "*kernel_stats.csv" | head -1) gpurun_out/r04/kernel_stats_radix_tier_64M.csv
rm -rf gpurun_out/r04/rxstats
head -8 gpurun_out/r04/kernel_stats_radix_tier_64M.csv | cut -c1-120
