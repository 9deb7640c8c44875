set -e
mkdir -p gpurun_out/r04
timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-other-configs > gpurun_out/r04/q1_sf100_quick.json 2> gpurun_out/r04/q1_sf100_quick.err || { tail -30 gpurun_out/r04/q1_sf100_quick.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r04/q1_sf100_quick.json"))
print("sf=100", round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["time_split_ms"].items()}, round(d["roofline"]["frac"],4), d["full_check"]["gpu_matches_oracle_full"], d["full_check"]["f32_ulp_flips_full"])
PY
start=$(date +%s)
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r04/gputest_full.log 2>&1 || { tail -80 gpurun_out/r04/gputest_full.log; exit 1; }
echo "gpu suite wall: $(( $(date +%s) - start )) s"
tail -32 gpurun_out/r04/gputest_full.log
for mode in plain rccl p2p; do
  unset HIPSPARK_FORCE_DIST HIPSPARK_P2P_SLABS
  if [ "$mode" = rccl ]; then export HIPSPARK_FORCE_DIST=1; fi
  if [ "$mode" = p2p ]; then export HIPSPARK_FORCE_DIST=1 HIPSPARK_P2P_SLABS=1; fi
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29777 timeout -k 10 200 python bench.py --sf 12.5 --steps 40 --no-cpu-baseline --no-other-configs > gpurun_out/r04/q1_sf12.5_$mode.json 2> gpurun_out/r04/q1_sf12.5_$mode.err || { tail -20 gpurun_out/r04/q1_sf12.5_$mode.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r04/q1_sf12.5_$mode.json"))
print("sf=12.5 $mode", round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["time_split_ms"].items()}, round(d["roofline"]["frac"],4))
PY
done
unset HIPSPARK_FORCE_DIST HIPSPARK_P2P_SLABS
HIPSPARK_DIST_BACKEND=gloo HIPSPARK_FORCE_DEVICE=0 timeout -k 10 400 python bench.py --config join --gpus 4 --steps 10 --no-cpu-baseline > gpurun_out/r04/join_4ranks_gloo.json 2> gpurun_out/r04/join_4ranks_gloo.err || { tail -30 gpurun_out/r04/join_4ranks_gloo.err; exit 1; }
HIPSPARK_DIST_BACKEND=gloo HIPSPARK_FORCE_DEVICE=0 HIPSPARK_SHARDED_BUILD=0 timeout -k 10 400 python bench.py --config join --gpus 4 --steps 10 --no-cpu-baseline > gpurun_out/r04/join_4ranks_gloo_gathered.json 2> gpurun_out/r04/join_4ranks_gloo_gathered.err || { tail -30 gpurun_out/r04/join_4ranks_gloo_gathered.err; exit 1; }
HIPSPARK_DIST_BACKEND=gloo HIPSPARK_FORCE_DEVICE=0 timeout -k 10 300 python bench.py --gpus 2 --sf 2 --steps 10 --no-cpu-baseline > gpurun_out/r04/q1_sf2_2ranks_selflaunch.json 2> gpurun_out/r04/q1_sf2_2ranks_selflaunch.err || { tail -30 gpurun_out/r04/q1_sf2_2ranks_selflaunch.err; exit 1; }
python - <<PY
import json
for f in ("join_4ranks_gloo","join_4ranks_gloo_gathered","q1_sf2_2ranks_selflaunch"):
    d=json.loads([l for l in open(f"gpurun_out/r04/{f}.json") if l.startswith("{")][-1])
    print(f, d["n_gpus"], round(d["ms_per_step"],3), {k:round(v,4) for k,v in d["time_split_ms"].items()}, d["full_check"]["gpu_matches_oracle_full"], d["config"].get("join"))
PY
