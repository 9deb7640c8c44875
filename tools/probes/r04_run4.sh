set -e
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_radix_tier.py tests/test_gpu_shared_tier.py -x -q --durations=8 > gpurun_out/r04/t_radix.log 2>&1 || { tail -60 gpurun_out/r04/t_radix.log; exit 1; }
tail -14 gpurun_out/r04/t_radix.log
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py -q --durations=12 > gpurun_out/r04/t_dist.log 2>&1 || { grep -n "rank .* job\|Error\|FAILED" gpurun_out/r04/t_dist.log | head -40; tail -5 gpurun_out/r04/t_dist.log; }
tail -20 gpurun_out/r04/t_dist.log
timeout -k 10 900 python bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err || { tail -40 gpurun_out/r04/bench_default.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r04/bench_default.json"))
print(d["ms_per_step"], d["time_split_ms"], d["roofline"]["frac"], d["cold"])
for k,v in d["other_configs"].items():
    print(k, round(v["ms_per_step"],4), {a:round(b,4) for a,b in v["time_split_ms"].items()}, round(v["roofline"]["frac"],3), v["full_check"]["gpu_matches_oracle_full"], v.get("cache_resident"), v["cold"])
PY
