set -e
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r04/prof_ops --output-format csv -- python3 tools/bench_ops.py > gpurun_out/r04/prof_ops.log 2>&1 || { tail -20 gpurun_out/r04/prof_ops.log; exit 1; }
f=$(find gpurun_out/r04/prof_ops -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/r04/ops_kernel_stats.csv
find gpurun_out/r04/prof_ops -name '*kernel_trace.csv' -delete
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/r04/ops_kernel_stats.csv")))
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("k_jh_","k_jd_","k_rx_","k_join_","k_scan","k_fill")):
        print(f'{n[:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}')
PY
