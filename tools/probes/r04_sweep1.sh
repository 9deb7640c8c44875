set -e
mkdir -p gpurun_out/r04
for cs in 0 24 32 40 48 64 80 98 128; do
  if [ "$cs" = 0 ]; then unset HIPSPARK_CHUNK_STEPS; else export HIPSPARK_CHUNK_STEPS=$cs; fi
  timeout -k 10 120 python bench.py --sf 12.5 --steps 40 --no-cpu-baseline --no-full-check --no-other-configs > gpurun_out/r04/sweep_cs_$cs.json 2> gpurun_out/r04/sweep_cs_$cs.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r04/sweep_cs_$cs.json"))
print("cs=$cs", round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["time_split_ms"].items()}, round(d["roofline"]["frac"],4), d["roofline"]["launch"]["chunks"], flush=True)
PY
done
unset HIPSPARK_CHUNK_STEPS
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=40 > gpurun_out/r04/gputest_start.log 2>&1
tail -60 gpurun_out/r04/gputest_start.log
