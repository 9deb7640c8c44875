set -e
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_radix_tier.py tests/test_gpu_kernels.py -m gpu -x -q > gpurun_out/r04/gputest_subset.log 2>&1 || { tail -60 gpurun_out/r04/gputest_subset.log; exit 1; }
tail -2 gpurun_out/r04/gputest_subset.log
( timeout -k 10 300 python tools/bench_radix.py 67108864 4194304 1 5; timeout -k 10 300 python tools/bench_radix.py 600037902 500000 287 3 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/radix_tier_try.txt
cat gpurun_out/r04/radix_tier_try.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/rxstats -- python3 tools/bench_radix.py 67108864 4194304 1 5 > /dev/null 2>&1
grep -E "k_rx_|k_jd|k_jh" $(find gpurun_out/r04/rxstats -name "*kernel_stats.csv" | head -1) | cut -c1-110
rm -rf gpurun_out/r04/rxstats
