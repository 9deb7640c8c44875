set -e
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q > gpurun_out/r04/gputest_subset.log 2>&1 || { tail -60 gpurun_out/r04/gputest_subset.log; exit 1; }
tail -2 gpurun_out/r04/gputest_subset.log
last_json() { python - "$1" <<'PY'
import json, sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1].split("/")[-1], d["n_gpus"], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["time_split_ms"].items()}, round(d["roofline"]["frac"],4), (d.get("full_check") or {}).get("gpu_matches_oracle_full"))
PY
}
for mode in rccl p2p rccl p2p; do
  unset HIPSPARK_FORCE_DIST HIPSPARK_P2P_SLABS
  if [ "$mode" = rccl ]; then export HIPSPARK_FORCE_DIST=1; fi
  if [ "$mode" = p2p ]; then export HIPSPARK_FORCE_DIST=1 HIPSPARK_P2P_SLABS=1; fi
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29777 timeout -k 10 200 python bench.py --sf 12.5 --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/r04/q1_sf12.5_$mode.json 2> gpurun_out/r04/q1_sf12.5_$mode.err || { tail -20 gpurun_out/r04/q1_sf12.5_$mode.err; exit 1; }
  last_json gpurun_out/r04/q1_sf12.5_$mode.json
done
unset HIPSPARK_FORCE_DIST HIPSPARK_P2P_SLABS
HIPSPARK_DIST_BACKEND=gloo HIPSPARK_FORCE_DEVICE=0 timeout -k 10 300 python bench.py --gpus 2 --sf 2 --steps 10 --no-cpu-baseline > gpurun_out/r04/q1_sf2_2ranks.json 2> gpurun_out/r04/q1_sf2_2ranks.err || { tail -30 gpurun_out/r04/q1_sf2_2ranks.err; exit 1; }
last_json gpurun_out/r04/q1_sf2_2ranks.json
