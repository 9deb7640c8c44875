set -e
[ -x tools/probes/read_ceiling ] || hipcc --offload-arch=gfx950 -O3 -o tools/probes/read_ceiling tools/probes/read_ceiling.hip
mkdir -p gpurun_out/r04
timeout -k 10 200 python tools/scan_stamps.py 12.5 > gpurun_out/r04/scan_stamps_sf12.5.txt 2> gpurun_out/r04/scan_stamps.err || { tail -20 gpurun_out/r04/scan_stamps.err; exit 1; }
cat gpurun_out/r04/scan_stamps_sf12.5.txt
timeout -k 10 200 python tools/scan_stamps.py 100 > gpurun_out/r04/scan_stamps_sf100.txt 2> gpurun_out/r04/scan_stamps.err || { tail -20 gpurun_out/r04/scan_stamps.err; exit 1; }
cat gpurun_out/r04/scan_stamps_sf100.txt
timeout -k 10 200 python tools/scan_stamps.py 1 > gpurun_out/r04/scan_stamps_sf1.txt 2> gpurun_out/r04/scan_stamps.err || { tail -20 gpurun_out/r04/scan_stamps.err; exit 1; }
cat gpurun_out/r04/scan_stamps_sf1.txt
timeout -k 10 300 tools/probes/read_ceiling > gpurun_out/r04/read_ceiling_sf100.txt 2>&1 || { tail -20 gpurun_out/r04/read_ceiling_sf100.txt; exit 1; }
cat gpurun_out/r04/read_ceiling_sf100.txt
timeout -k 10 300 tools/probes/read_ceiling 75004738 > gpurun_out/r04/read_ceiling_sf12.5.txt 2>&1 || { tail -20 gpurun_out/r04/read_ceiling_sf12.5.txt; exit 1; }
cat gpurun_out/r04/read_ceiling_sf12.5.txt
