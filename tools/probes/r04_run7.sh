set -e
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q > gpurun_out/r04/t_kern.log 2>&1 || { tail -60 gpurun_out/r04/t_kern.log; exit 1; }
tail -3 gpurun_out/r04/t_kern.log
timeout -k 10 300 python tools/bench_ops.py > gpurun_out/r04/ops.txt 2>&1 || { tail -30 gpurun_out/r04/ops.txt; exit 1; }
grep "A1\|A3 hs_compact\|A8" gpurun_out/r04/ops.txt
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04/prof_ops2 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_ops.py > $GRAFT_REPO_ROOT/gpurun_out/r04/ops_prof.txt 2>&1 || true
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r04/prof_ops2/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if any(t in r["Name"] for t in ("k_jd", "k_rx_scatter4<1", "k_rx_hist4", "k_rx_next", "k_rx_tiles", "k_mask", "k_lens")):
            print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
timeout -k 10 900 python -m pytest tests/test_gpu_join_dict.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_stage_abi.py tests/test_gpu_streaming.py -x -q > gpurun_out/r04/t_more.log 2>&1 || { tail -60 gpurun_out/r04/t_more.log; exit 1; }
tail -4 gpurun_out/r04/t_more.log
