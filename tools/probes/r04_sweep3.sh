set -e
mkdir -p gpurun_out/r04
for sf in 1 3; do
for cs in 0 8 10 12 14 16 20 24; do
  if [ $cs = 0 ]; then unset HIPSPARK_CHUNK_STEPS; else export HIPSPARK_CHUNK_STEPS=$cs; fi
  timeout -k 10 200 python bench.py --sf $sf --steps 60 --no-cpu-baseline --no-full-check --no-other-configs > gpurun_out/r04/sw.json 2> gpurun_out/r04/sw.err || { tail -5 gpurun_out/r04/sw.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r04/sw.json") if l.startswith("{")][-1])
print("sf=$sf cs=$cs", round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["time_split_ms"].items()}, d["roofline"]["launch"]["chunks"])
PY
done
done
