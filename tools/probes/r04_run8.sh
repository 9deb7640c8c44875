set -e
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "compact or offsets or scans" > gpurun_out/r04/t_kern.log 2>&1 || { tail -60 gpurun_out/r04/t_kern.log; exit 1; }
tail -3 gpurun_out/r04/t_kern.log
timeout -k 10 300 python tools/bench_ops.py > gpurun_out/r04/ops.txt 2>&1 || { tail -30 gpurun_out/r04/ops.txt; exit 1; }
grep "A1\|A3 hs_compact" gpurun_out/r04/ops.txt
bash tools/probes/r04_sweep2.sh
