# round-4 evidence: bench lines, kernel stats, PMC traffic passes (run from the repo root on the GPU box)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04ev
mkdir -p $O
cd $R
timeout -k 10 500 python bench.py > $O/r04_bench_q1_sf100.json 2> $O/bench_default.err || { tail -30 $O/bench_default.err; exit 1; }
echo "bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/q1stats -- python3 $R/bench.py --no-cpu-baseline --no-full-check --no-other-configs > $O/r04_bench_q1_sf100_under_rocprof.json 2> $O/q1stats.log
find $O/q1stats -name '*kernel_trace.csv' -delete
cp $(find $O/q1stats -name "*kernel_stats.csv" | head -1) $O/r04_kernel_stats_q1_sf100.csv
echo "stats done"
for cfg in q1 join strkey; do
  extra=""; if [ $cfg != q1 ]; then extra="--config $cfg"; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc_${ctr}_$cfg -- python3 $R/bench.py $extra --steps 3 --warmup 1 --no-cpu-baseline --no-full-check --no-other-configs > $O/pmc_${ctr}_$cfg.log 2>&1 || { tail -20 $O/pmc_${ctr}_$cfg.log; exit 1; }
    find $O/pmc_${ctr}_$cfg -name '*kernel_trace.csv' -delete
  done
  echo "pmc $cfg done"
done
cd $R
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE_q1 $O/pmc_WRITE_SIZE_q1 q1 600037902 > $O/r04_pmc_hbm_traffic_q1_sf100.json
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE_join $O/pmc_WRITE_SIZE_join join 59986052 > $O/r04_pmc_hbm_traffic_join_sf10.json
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE_strkey $O/pmc_WRITE_SIZE_strkey strkey 59986052 > $O/r04_pmc_hbm_traffic_strkey_sf10.json
python tools/pmc_table.py $O/pmc_FETCH_SIZE_join $O/pmc_WRITE_SIZE_join > $O/r04_pmc_config4_join_sf10.txt
head -12 $O/r04_pmc_hbm_traffic_q1_sf100.json
# the generic operators under the counters (the join forms above all)
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc_${ctr}_ops -- python3 $R/tools/bench_ops.py > $O/pmc_${ctr}_ops.log 2>&1 || { tail -20 $O/pmc_${ctr}_ops.log; exit 1; }
  find $O/pmc_${ctr}_ops -name '*kernel_trace.csv' -delete
done
cd $R
python tools/pmc_table.py $O/pmc_FETCH_SIZE_ops $O/pmc_WRITE_SIZE_ops > $O/r04_pmc_ops_64M.txt
grep -E "k_jh_|k_jd_|k_join_|k_rx_scatter4|k_rx_hist4" $O/r04_pmc_ops_64M.txt
# counter folders are large: keep the summaries only
rm -rf $O/pmc_FETCH_SIZE_* $O/pmc_WRITE_SIZE_* $O/q1stats
for mode in plain rccl p2p; do
  unset HIPSPARK_FORCE_DIST HIPSPARK_P2P_SLABS
  if [ "$mode" = rccl ]; then export HIPSPARK_FORCE_DIST=1; fi
  if [ "$mode" = p2p ]; then export HIPSPARK_FORCE_DIST=1 HIPSPARK_P2P_SLABS=1; fi
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29777 timeout -k 10 200 python bench.py --sf 12.5 --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs > $O/r04_bench_q1_sf12.5_$mode.json 2> $O/q1_sf12.5_$mode.err || { tail -20 $O/q1_sf12.5_$mode.err; exit 1; }
done
unset HIPSPARK_FORCE_DIST HIPSPARK_P2P_SLABS
HIPSPARK_DIST_BACKEND=gloo HIPSPARK_FORCE_DEVICE=0 timeout -k 10 400 python bench.py --config join --gpus 4 --steps 10 --no-cpu-baseline > $O/r04_join_4ranks_gloo.json 2> $O/join_4ranks_gloo.err || { tail -30 $O/join_4ranks_gloo.err; exit 1; }
HIPSPARK_DIST_BACKEND=gloo HIPSPARK_FORCE_DEVICE=0 timeout -k 10 300 python bench.py --gpus 2 --sf 2 --steps 10 --no-cpu-baseline > $O/r04_q1_sf2_2ranks_selflaunch.json 2> $O/q1_sf2_2ranks.err || { tail -30 $O/q1_sf2_2ranks.err; exit 1; }
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/r04_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    if "ms_per_step" in d:
        print(f.split("/")[-1], d["n_gpus"], round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["time_split_ms"].items()}, round(d["roofline"]["frac"],4), d["roofline"].get("traffic"), (d.get("full_check") or {}).get("gpu_matches_oracle_full"))
        for k,v in (d.get("other_configs") or {}).items():
            print("   ", k, round(v["ms_per_step"],4), {a:round(b,4) for a,b in v["time_split_ms"].items()}, round(v["roofline"]["frac"],4), v["full_check"]["gpu_matches_oracle_full"])
PY
