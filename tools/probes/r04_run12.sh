set -e
mkdir -p gpurun_out/r04
last_json() { python - "$1" <<'PY'
import json, sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1].split("/")[-1], d["n_gpus"], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["time_split_ms"].items()}, round(d["roofline"]["frac"],4), (d.get("full_check") or {}).get("gpu_matches_oracle_full"), d["config"].get("join"))
PY
}
for rep in 1 2; do
for form in late early; do
  unset HIPSPARK_JIT_EARLY_CUT
  if [ "$form" = early ]; then export HIPSPARK_JIT_EARLY_CUT=1; fi
  for sf in 100 12.5; do
    timeout -k 10 300 python bench.py --sf $sf --steps 30 --no-cpu-baseline --no-other-configs --no-full-check > gpurun_out/r04/ab_${form}_sf${sf}_$rep.json 2> gpurun_out/r04/ab.err || { tail -30 gpurun_out/r04/ab.err; exit 1; }
    last_json gpurun_out/r04/ab_${form}_sf${sf}_$rep.json
  done
done
done
unset HIPSPARK_JIT_EARLY_CUT
for mode in rccl p2p; do
  unset HIPSPARK_FORCE_DIST HIPSPARK_P2P_SLABS
  if [ "$mode" = rccl ]; then export HIPSPARK_FORCE_DIST=1; fi
  if [ "$mode" = p2p ]; then export HIPSPARK_FORCE_DIST=1 HIPSPARK_P2P_SLABS=1; fi
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29777 timeout -k 10 200 python bench.py --sf 12.5 --steps 40 --no-cpu-baseline --no-other-configs > gpurun_out/r04/q1_sf12.5_$mode.json 2> gpurun_out/r04/q1_sf12.5_$mode.err || { tail -20 gpurun_out/r04/q1_sf12.5_$mode.err; exit 1; }
  last_json gpurun_out/r04/q1_sf12.5_$mode.json
done
unset HIPSPARK_FORCE_DIST HIPSPARK_P2P_SLABS
HIPSPARK_DIST_BACKEND=gloo HIPSPARK_FORCE_DEVICE=0 timeout -k 10 400 python bench.py --config join --gpus 4 --steps 10 --no-cpu-baseline > gpurun_out/r04/join_4ranks_gloo.json 2> gpurun_out/r04/join_4ranks_gloo.err || { tail -30 gpurun_out/r04/join_4ranks_gloo.err; exit 1; }
last_json gpurun_out/r04/join_4ranks_gloo.json
HIPSPARK_DIST_BACKEND=gloo HIPSPARK_FORCE_DEVICE=0 HIPSPARK_SHARDED_BUILD=0 timeout -k 10 400 python bench.py --config join --gpus 4 --steps 10 --no-cpu-baseline > gpurun_out/r04/join_4ranks_gloo_gathered.json 2> gpurun_out/r04/join_4ranks_gloo_gathered.err || { tail -30 gpurun_out/r04/join_4ranks_gloo_gathered.err; exit 1; }
last_json gpurun_out/r04/join_4ranks_gloo_gathered.json
HIPSPARK_DIST_BACKEND=gloo HIPSPARK_FORCE_DEVICE=0 timeout -k 10 300 python bench.py --gpus 2 --sf 2 --steps 10 --no-cpu-baseline > gpurun_out/r04/q1_sf2_2ranks_selflaunch.json 2> gpurun_out/r04/q1_sf2_2ranks_selflaunch.err || { tail -30 gpurun_out/r04/q1_sf2_2ranks_selflaunch.err; exit 1; }
last_json gpurun_out/r04/q1_sf2_2ranks_selflaunch.json
