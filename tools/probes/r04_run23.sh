set -e
mkdir -p gpurun_out/r04
start=$(date +%s)
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r04/gputest_full.log 2>&1 || { tail -80 gpurun_out/r04/gputest_full.log; exit 1; }
echo "gpu suite wall: $(( $(date +%s) - start )) s"
tail -22 gpurun_out/r04/gputest_full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
