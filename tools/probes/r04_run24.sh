set -e
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for sf in 1 12.5; do
rm -rf gpurun_out/r04/tl_$sf
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04/tl_$sf -- python3 bench.py --sf $sf --steps 30 --no-cpu-baseline --no-other-configs --no-full-check > gpurun_out/r04/tl_$sf.log 2>&1 || { tail -20 gpurun_out/r04/tl_$sf.log; exit 1; }
f=$(find gpurun_out/r04/tl_$sf -name '*kernel_trace.csv' | head -1)
echo "sf=$sf"; python tools/trace_tail.py $f 8
rm -rf gpurun_out/r04/tl_$sf
done
for c in join strkey; do
rm -rf gpurun_out/r04/tl_$c
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04/tl_$c -- python3 bench.py --config $c --steps 30 --no-cpu-baseline --no-full-check > gpurun_out/r04/tl_$c.log 2>&1 || { tail -20 gpurun_out/r04/tl_$c.log; exit 1; }
f=$(find gpurun_out/r04/tl_$c -name '*kernel_trace.csv' | head -1)
echo "config $c"; python tools/trace_tail.py $f 26
rm -rf gpurun_out/r04/tl_$c
done
