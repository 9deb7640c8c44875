set -e
mkdir -p gpurun_out/r04
last_json() { python - "$1" <<'PY'
import json, sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1].split("/")[-1], d["n_gpus"], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["time_split_ms"].items()}, round(d["roofline"]["frac"],4), (d.get("full_check") or {}).get("gpu_matches_oracle_full"))
PY
}
for sf in 1 12.5 100; do
  timeout -k 10 300 python bench.py --sf $sf --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/r04/q1_sf${sf}_after.json 2> gpurun_out/r04/q1_after.err || { tail -30 gpurun_out/r04/q1_after.err; exit 1; }
  last_json gpurun_out/r04/q1_sf${sf}_after.json
done
for c in join strkey; do
  timeout -k 10 300 python bench.py --config $c --steps 40 --no-cpu-baseline > gpurun_out/r04/${c}_after.json 2> gpurun_out/r04/q1_after.err || { tail -30 gpurun_out/r04/q1_after.err; exit 1; }
  last_json gpurun_out/r04/${c}_after.json
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for sf in 1; do
rm -rf gpurun_out/r04/tl_$sf
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04/tl_$sf -- python3 bench.py --sf $sf --steps 30 --no-cpu-baseline --no-other-configs --no-full-check > gpurun_out/r04/tl_$sf.log 2>&1 || { tail -20 gpurun_out/r04/tl_$sf.log; exit 1; }
f=$(find gpurun_out/r04/tl_$sf -name '*kernel_trace.csv' | head -1)
echo "sf=$sf"; python tools/trace_tail.py $f 6
tail -1 gpurun_out/r04/tl_$sf.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('under rocprof: events say', d['roofline']['kernel_ms'])"
rm -rf gpurun_out/r04/tl_$sf
done
timeout -k 10 600 python -m pytest tests/test_gpu_q1_large.py tests/test_gpu_parity.py tests/test_gpu_distributed.py -m gpu -x -q 2>&1 | tail -3
