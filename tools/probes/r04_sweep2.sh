set -e
mkdir -p gpurun_out/r04
for sf in 1 3; do
for cs in 0 12 16 24 32 48 64 96; do
  if [ "$cs" = 0 ]; then unset HIPSPARK_CHUNK_STEPS; else export HIPSPARK_CHUNK_STEPS=$cs; fi
  timeout -k 10 120 python bench.py --sf $sf --steps 60 --no-cpu-baseline --no-full-check --no-other-configs > gpurun_out/r04/sweep_sf${sf}_cs_$cs.json 2> gpurun_out/r04/sweep_sf${sf}_cs_$cs.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r04/sweep_sf${sf}_cs_$cs.json"))
print("sf=$sf cs=$cs", round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["time_split_ms"].items()}, d["roofline"]["launch"]["chunks"], flush=True)
PY
done
done
