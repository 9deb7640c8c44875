set -e
mkdir -p gpurun_out/r04
for sf in 12.5 100 1; do
HIPSPARK_FINISH_STAMPS=1 python tools/finish_phases.py $sf > gpurun_out/r04/finish_phases_sf$sf.txt 2>&1 || true
echo "sf=$sf"; grep -E "fold|total" gpurun_out/r04/finish_phases_sf$sf.txt
done
timeout -k 10 600 python -m pytest tests/test_gpu_ordered_fold.py tests/test_gpu_parity.py tests/test_gpu_q1_fullsize.py -m gpu -x -q 2>&1 | tail -3
