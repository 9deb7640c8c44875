# round-3 evidence, second batch: bench lines, finish phases, radix tier (run from the repo root on the GPU box)
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 500 python bench.py 2> gpurun_out/bench_default.err | tail -1 > gpurun_out/r03_bench_q1_sf100.json && \
timeout -k 10 300 python bench.py --sf 12.5 --steps 200 --warmup 10 --no-other-configs 2> /dev/null | tail -1 > gpurun_out/r03_bench_q1_sf12.5.json && \
HIPSPARK_FINISH_STAMPS=1 timeout -k 10 300 python tools/finish_phases.py 100 > gpurun_out/r03_finish_kernel_phases_sf100.txt 2>&1 && \
timeout -k 10 300 python bench.py --config strkey 2> /dev/null | tail -1 > gpurun_out/r03_bench_config5_strkey_sf10.json && \
( timeout -k 10 300 python tools/bench_radix.py 67108864 4194304 1 5; RADIX_BENCH_COUNT=1 timeout -k 10 300 python tools/bench_radix.py 67108864 4194304 1 4; RADIX_BENCH_STR=12 timeout -k 10 300 python tools/bench_radix.py 67108864 4194304 1 4; timeout -k 10 300 python tools/bench_radix.py 600037902 500000 287 3 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_radix_tier_64M.txt && \
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rxstats -- python3 $R/tools/bench_radix.py 67108864 4194304 1 5 > /dev/null 2>&1 && \
cd $R && cp $(find gpurun_out/rxstats -name "*kernel_stats.csv" | head -1) gpurun_out/r03_kernel_stats_radix_tier_64M.csv
python - <<PY
import json
for f in ("r03_bench_q1_sf100","r03_bench_q1_sf12.5","r03_bench_config5_strkey_sf10"):
    try:
        d=json.load(open(f"gpurun_out/{f}.json"))
        print(f, d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("traffic"), d.get("time_split_ms"), {k:(v.get("ms_per_step"), v.get("full_check",{}).get("gpu_matches_oracle_full")) for k,v in d.get("other_configs",{}).items()}, d.get("full_check",{}).get("gpu_matches_oracle_full"))
    except Exception as e: print(f, "FAILED", e)
PY
tail -12 gpurun_out/r03_finish_kernel_phases_sf100.txt; cat gpurun_out/r03_radix_tier_64M.txt | grep "run [234]"
