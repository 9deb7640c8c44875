set -e
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_ordered_fold.py tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_gpu_q1_fullsize.py tests/test_gpu_shared_tier.py tests/test_gpu_fuzz.py tests/test_gpu_q1_large.py tests/test_gpu_stage_abi.py tests/test_gpu_join_dict.py tests/test_gpu_sql.py tests/test_gpu_distributed.py -m gpu -x -q > gpurun_out/r04/gputest_subset.log 2>&1 || { tail -80 gpurun_out/r04/gputest_subset.log; exit 1; }
tail -3 gpurun_out/r04/gputest_subset.log
for sf in 12.5 100 1; do
HIPSPARK_FINISH_STAMPS=1 python tools/finish_phases.py $sf > gpurun_out/r04/finish_phases_sf$sf.txt 2>&1 || true
echo "sf=$sf"; grep -v amdgpu gpurun_out/r04/finish_phases_sf$sf.txt
done
last_json() { python - "$1" <<'PY'
import json, sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1].split("/")[-1], d["n_gpus"], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["time_split_ms"].items()}, round(d["roofline"]["frac"],4), (d.get("full_check") or {}).get("gpu_matches_oracle_full"))
PY
}
for sf in 12.5 1; do
  timeout -k 10 300 python bench.py --sf $sf --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/r04/q1_sf${sf}_after.json 2> gpurun_out/r04/q1_after.err || { tail -30 gpurun_out/r04/q1_after.err; exit 1; }
  last_json gpurun_out/r04/q1_sf${sf}_after.json
done
for c in join strkey; do
  timeout -k 10 300 python bench.py --config $c --steps 40 --no-cpu-baseline > gpurun_out/r04/${c}_after.json 2> gpurun_out/r04/q1_after.err || { tail -30 gpurun_out/r04/q1_after.err; exit 1; }
  last_json gpurun_out/r04/${c}_after.json
done
