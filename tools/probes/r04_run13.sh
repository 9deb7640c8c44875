set -e
mkdir -p gpurun_out/r04
start=$(date +%s)
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=30 > gpurun_out/r04/gputest_full.log 2>&1 || { tail -80 gpurun_out/r04/gputest_full.log; exit 1; }
echo "gpu suite wall: $(( $(date +%s) - start )) s"
tail -36 gpurun_out/r04/gputest_full.log
start=$(date +%s)
timeout -k 10 500 python bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err || { tail -30 gpurun_out/r04/bench_default.err; exit 1; }
echo "default bench wall: $(( $(date +%s) - start )) s"
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r04/bench_default.json") if l.startswith("{")][-1])
print("sf=100", round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["time_split_ms"].items()}, round(d["roofline"]["frac"],4), d["full_check"]["gpu_matches_oracle_full"], d["roofline"].get("traffic_source"))
print("cold", d["cold"])
for k,v in d["other_configs"].items():
    print(k, round(v["ms_per_step"],4), {a:round(b,4) for a,b in v["time_split_ms"].items()}, round(v["roofline"]["frac"],4), v["full_check"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
for mode in plain rccl; do
  unset HIPSPARK_FORCE_DIST
  if [ "$mode" = rccl ]; then export HIPSPARK_FORCE_DIST=1; fi
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29777 timeout -k 10 200 python bench.py --sf 12.5 --steps 40 --no-cpu-baseline --no-other-configs > gpurun_out/r04/q1_sf12.5_$mode.json 2> gpurun_out/r04/q1_sf12.5_$mode.err || { tail -20 gpurun_out/r04/q1_sf12.5_$mode.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r04/q1_sf12.5_$mode.json") if l.startswith("{")][-1])
print("sf=12.5 $mode", round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["time_split_ms"].items()}, round(d["roofline"]["frac"],4))
PY
done
unset HIPSPARK_FORCE_DIST
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r04/prof_q1 --output-format csv -- python3 bench.py --no-cpu-baseline --no-other-configs --no-full-check > gpurun_out/r04/prof_q1.log 2>&1 || { tail -20 gpurun_out/r04/prof_q1.log; exit 1; }
find gpurun_out/r04/prof_q1 -name '*kernel_stats.csv' | head -3
find gpurun_out/r04/prof_q1 -name '*kernel_trace.csv' -delete
python tools/bench_ops.py > gpurun_out/r04/ops_microbench.txt 2> gpurun_out/r04/ops_microbench.err || { tail -20 gpurun_out/r04/ops_microbench.err; exit 1; }
cat gpurun_out/r04/ops_microbench.txt
