set -e
mkdir -p gpurun_out/r04
for sf in 12.5 1; do
timeout -k 10 200 python tools/scan_stamps.py $sf > gpurun_out/r04/scan_stamps_after_sf$sf.txt 2> gpurun_out/r04/scan_stamps.err || { tail -20 gpurun_out/r04/scan_stamps.err; exit 1; }
head -24 gpurun_out/r04/scan_stamps_after_sf$sf.txt
done
