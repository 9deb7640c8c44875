# round-3 evidence for the Q1 line: separate PMC passes, kernel stats (run from the repo root on the GPU box)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-full-check --no-other-configs > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-full-check --no-other-configs > $R/gpurun_out/pmc_write.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/q1stats -- python3 $R/bench.py --no-cpu-baseline --no-full-check --no-other-configs > $R/gpurun_out/r03_bench_q1_sf100_under_rocprof.json 2> $R/gpurun_out/q1stats.log
cd $R
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 600037902 > gpurun_out/r03_pmc_hbm_traffic_q1_sf100.json
cat gpurun_out/r03_pmc_hbm_traffic_q1_sf100.json
cp $(find gpurun_out/q1stats -name "*kernel_stats.csv" | head -1) gpurun_out/r03_kernel_stats_q1_sf100.csv
head -5 gpurun_out/r03_kernel_stats_q1_sf100.csv | cut -c1-150
tail -1 gpurun_out/r03_bench_q1_sf100_under_rocprof.json | cut -c1-400
