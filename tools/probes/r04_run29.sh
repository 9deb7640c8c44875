set -e
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_sql.py tests/test_gpu_streaming.py -m gpu -x -q > gpurun_out/r04/gputest_subset.log 2>&1 || { tail -60 gpurun_out/r04/gputest_subset.log; exit 1; }
tail -2 gpurun_out/r04/gputest_subset.log
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r04/prof_ops --output-format csv -- python3 tools/bench_ops.py > gpurun_out/r04/ops_microbench.txt 2> gpurun_out/r04/prof_ops.log || { tail -20 gpurun_out/r04/prof_ops.log; exit 1; }
grep "A8\|A5" gpurun_out/r04/ops_microbench.txt
f=$(find gpurun_out/r04/prof_ops -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/r04/ops_kernel_stats.csv
rm -rf gpurun_out/r04/prof_ops
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/r04/ops_kernel_stats.csv")):
    n=r["Name"]
    if any(k in n for k in ("k_jh_","k_jd_")):
        print(f'{n[:40]:40s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}')
PY
