set -e
mkdir -p gpurun_out/r04
HIPSPARK_DIST_FUZZ=10:70 HIPSPARK_DIST_WIDE=4:30 timeout -k 10 1100 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q > gpurun_out/r04/fuzz_e.log 2>&1 || { tail -60 gpurun_out/r04/fuzz_e.log; exit 1; }
tail -2 gpurun_out/r04/fuzz_e.log
HIPSPARK_P2P_SLABS=1 timeout -k 10 600 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q -k "slab or rccl or bench" > gpurun_out/r04/fuzz_f.log 2>&1 || { tail -60 gpurun_out/r04/fuzz_f.log; exit 1; }
tail -2 gpurun_out/r04/fuzz_f.log
