# a wider hunt than the default sample of the randomised tests (round 4, after the scan epilogue / unit combine / finish changes)
set -e
mkdir -p gpurun_out/r04
HIPSPARK_FUZZ_FIRST=48 HIPSPARK_FUZZ_SEEDS=400 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r04/fuzz_a.log 2>&1 || { tail -60 gpurun_out/r04/fuzz_a.log; exit 1; }
tail -2 gpurun_out/r04/fuzz_a.log
HIPSPARK_WIDE_FIRST=12 HIPSPARK_WIDE_SEEDS=150 timeout -k 10 900 python -m pytest tests/test_gpu_shared_tier.py -m gpu -x -q > gpurun_out/r04/fuzz_b.log 2>&1 || { tail -60 gpurun_out/r04/fuzz_b.log; exit 1; }
tail -2 gpurun_out/r04/fuzz_b.log
HIPSPARK_HC_FIRST=8 HIPSPARK_HC_SEEDS=80 HIPSPARK_HS_FIRST=6 HIPSPARK_HS_SEEDS=40 timeout -k 10 900 python -m pytest tests/test_gpu_radix_tier.py -m gpu -x -q > gpurun_out/r04/fuzz_c.log 2>&1 || { tail -60 gpurun_out/r04/fuzz_c.log; exit 1; }
tail -2 gpurun_out/r04/fuzz_c.log
timeout -k 10 900 python tools/probes/stage_fuzz.py 1000 400 > gpurun_out/r04/fuzz_d.log 2>&1 || { tail -40 gpurun_out/r04/fuzz_d.log; exit 1; }
tail -3 gpurun_out/r04/fuzz_d.log
