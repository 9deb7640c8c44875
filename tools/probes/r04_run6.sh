set -e
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_join_shard.py -x -q > gpurun_out/r04/t_kern.log 2>&1 || { tail -60 gpurun_out/r04/t_kern.log; exit 1; }
tail -3 gpurun_out/r04/t_kern.log
timeout -k 10 300 python tools/bench_ops.py > gpurun_out/r04/ops.txt 2>&1 || { tail -30 gpurun_out/r04/ops.txt; exit 1; }
cat gpurun_out/r04/ops.txt
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04/prof_ops --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_ops.py 16777216 > $GRAFT_REPO_ROOT/gpurun_out/r04/ops_prof.txt 2>&1 || true
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r04/prof_ops/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if any(t in r["Name"] for t in ("k_jd", "k_rx_", "k_scan", "k_join", "k_bytes", "k_mask", "k_lens", "k_part")):
            print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_join_dict.py tests/test_gpu_parity.py -x -q -k "config4 or join or golden" > gpurun_out/r04/t_dist.log 2>&1 || { grep -n "rank .* job\|Error\|FAILED" gpurun_out/r04/t_dist.log | head -40; tail -5 gpurun_out/r04/t_dist.log; exit 1; }
tail -4 gpurun_out/r04/t_dist.log
