set -e
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_streaming.py tests/test_gpu_q1_large.py tests/test_gpu_kernels.py -x -q > gpurun_out/r04/t_stream.log 2>&1 || { tail -80 gpurun_out/r04/t_stream.log; exit 1; }
tail -3 gpurun_out/r04/t_stream.log
timeout -k 10 300 python bench.py --sf 1 --steps 60 --no-cpu-baseline --no-full-check --no-other-configs > gpurun_out/r04/sf1_after_floor.json 2>gpurun_out/r04/sf1_after_floor.err
python - <<PY
import json
d=json.load(open("gpurun_out/r04/sf1_after_floor.json"))
print("sf=1", round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["time_split_ms"].items()}, d["roofline"]["launch"]["chunks"])
PY
