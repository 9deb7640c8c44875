set -e
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_gpu_q1_fullsize.py tests/test_gpu_shared_tier.py tests/test_gpu_join_dict.py tests/test_gpu_fuzz.py tests/test_gpu_q1_large.py -m gpu -x -q > gpurun_out/r04/gputest_subset.log 2>&1 || { tail -60 gpurun_out/r04/gputest_subset.log; exit 1; }
tail -3 gpurun_out/r04/gputest_subset.log
for sf in 12.5 1 100; do
timeout -k 10 200 python tools/scan_stamps.py $sf > gpurun_out/r04/scan_stamps_after_sf$sf.txt 2> gpurun_out/r04/scan_stamps.err || { tail -20 gpurun_out/r04/scan_stamps.err; exit 1; }
head -16 gpurun_out/r04/scan_stamps_after_sf$sf.txt
done
last_json() { python - "$1" <<'PY'
import json, sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1].split("/")[-1], d["n_gpus"], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["time_split_ms"].items()}, round(d["roofline"]["frac"],4), (d.get("full_check") or {}).get("gpu_matches_oracle_full"))
PY
}
for sf in 12.5 1 100; do
  timeout -k 10 300 python bench.py --sf $sf --steps 40 --no-cpu-baseline --no-other-configs > gpurun_out/r04/q1_sf${sf}_after.json 2> gpurun_out/r04/q1_after.err || { tail -30 gpurun_out/r04/q1_after.err; exit 1; }
  last_json gpurun_out/r04/q1_sf${sf}_after.json
done
export HIPSPARK_FORCE_DIST=1
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29777 timeout -k 10 200 python bench.py --sf 12.5 --steps 40 --no-cpu-baseline --no-other-configs > gpurun_out/r04/q1_sf12.5_rccl_after.json 2> gpurun_out/r04/q1_after.err || { tail -20 gpurun_out/r04/q1_after.err; exit 1; }
last_json gpurun_out/r04/q1_sf12.5_rccl_after.json
unset HIPSPARK_FORCE_DIST
for c in join strkey; do
  timeout -k 10 300 python bench.py --config $c --steps 40 --no-cpu-baseline > gpurun_out/r04/${c}_after.json 2> gpurun_out/r04/q1_after.err || { tail -30 gpurun_out/r04/q1_after.err; exit 1; }
  last_json gpurun_out/r04/${c}_after.json
done
