"""HBM traffic of the scan kernel per launch from two rocprofv3 counter passes (collected separately, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes):
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write <rows per launch> > profiles/rNN_pmc_hbm_traffic_q1_sf100.json
FETCH_SIZE / WRITE_SIZE count KB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream (guide,
section HBM), hence x2 on the read side."""
import csv, glob, json, subprocess, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import kernel_sources_sha

def avg_counter(folder, counter, kernel_prefix="k_agg_jit"):
    vals = []
    for path in glob.glob(f"{folder}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and r["Kernel_Name"].startswith(kernel_prefix):
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for {kernel_prefix} under {folder}")
    return sum(vals) / len(vals), len(vals)

fetch_kb, nf = avg_counter(sys.argv[1], "FETCH_SIZE")
write_kb, nw = avg_counter(sys.argv[2], "WRITE_SIZE")
rows = int(sys.argv[3])
algo = 26 * rows
read_b, write_b = fetch_kb * 1024 * 2, write_kb * 1024
print(json.dumps({
    "kernel": f"k_agg_jit (Q1, {rows} rows per launch)", "launches_averaged": [nf, nw],
    "FETCH_SIZE_KB_avg": fetch_kb, "hbm_read_bytes_per_launch (FETCH_SIZE*1024*2)": read_b,
    "WRITE_SIZE_KB_avg": write_kb, "hbm_write_bytes_per_launch": write_b,
    "algorithmic_bytes_per_launch (26 B/row)": algo, "traffic_over_algorithmic": (read_b + write_b) / algo,
    # bench.py quotes these figures only while the device code is the code they were measured on
    "kernel_sources_sha": kernel_sources_sha(),
    "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "?",
}, indent=1))
