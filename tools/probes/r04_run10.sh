set -e
mkdir -p gpurun_out/r04
start=$(date +%s)
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r04/gputest_full.log 2>&1 || { tail -80 gpurun_out/r04/gputest_full.log; exit 1; }
echo "gpu suite wall: $(( $(date +%s) - start )) s"
tail -40 gpurun_out/r04/gputest_full.log
timeout -k 10 900 python bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err || { tail -40 gpurun_out/r04/bench_default.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r04/bench_default.json"))
print(d["ms_per_step"], d["time_split_ms"], d["roofline"]["frac"], d["cold"])
for k,v in d["other_configs"].items():
    print(k, round(v["ms_per_step"],4), {a:round(b,4) for a,b in v["time_split_ms"].items()}, round(v["roofline"]["frac"],3), v["full_check"]["gpu_matches_oracle_full"], v.get("cache_resident"), v["cold"])
PY
