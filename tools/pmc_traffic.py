"""HBM traffic per launch of a workload's dominant kernel from two rocprofv3 counter passes (collected separately, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes; CONFIG = q1 | join | strkey, the bench.py --config of the same name):

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_CONFIG --output-format csv -- python3 bench.py [--config CONFIG] --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write_CONFIG --output-format csv -- python3 bench.py [--config CONFIG] --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs
    python tools/pmc_traffic.py gpurun_out/pmc_fetch_CONFIG gpurun_out/pmc_write_CONFIG CONFIG <rows per launch> > profiles/rNN_pmc_hbm_traffic_<...>.json

FETCH_SIZE / WRITE_SIZE count KB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream (guide, section
HBM), hence x2 on the read side.  The file carries the hash of the kernel sources it was measured on: bench.py quotes it
only while that hash is current."""
import csv, glob, json, subprocess, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import TRAFFIC_KERNELS, kernel_sources_sha


def per_kernel(folder, counter):
    """-> {kernel name: (average counter value over its launches, launches)}"""
    acc = {}
    for path in glob.glob(f"{folder}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                name = r["Kernel_Name"].split("(")[0]
                acc.setdefault(name, []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
config, rows = sys.argv[3], int(sys.argv[4])
spec = TRAFFIC_KERNELS[config]
dominant = next((k for k in fetch if k.startswith(spec["dominant"])), None)
if dominant is None:
    raise SystemExit(f"no FETCH_SIZE rows for {spec['dominant']} under {sys.argv[1]}: {sorted(fetch)}")
fetch_kb, nf = fetch[dominant]
write_kb, nw = write.get(dominant, (0.0, 0))
read_b, write_b = fetch_kb * 1024 * 2, write_kb * 1024
out = {
    "config": config, "kernel": f"{dominant} ({rows} rows per launch)", "launches_averaged": [nf, nw],
    "FETCH_SIZE_KB_avg": fetch_kb, "hbm_read_bytes_per_launch (FETCH_SIZE*1024*2)": read_b,
    "WRITE_SIZE_KB_avg": write_kb, "hbm_write_bytes_per_launch": write_b,
    # every other kernel of a step, for the split lines of the bench (the join's table build)
    "other_kernels": {k: {"hbm_read_bytes (FETCH_SIZE*1024*2)": fetch[k][0] * 1024 * 2, "hbm_write_bytes": write.get(k, (0.0, 0))[0] * 1024,
                          "launches": fetch[k][1]}
                      for k in sorted(fetch) if k != dominant and any(k.startswith(p) for p in spec["others"])},
    # bench.py quotes these figures only while the device code is the code they were measured on
    "kernel_sources_sha": kernel_sources_sha(spec["sources"]),
    "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "?",
}
if config == "q1":
    out["algorithmic_bytes_per_launch (26 B/row)"] = 26 * rows
    out["traffic_over_algorithmic"] = (read_b + write_b) / (26 * rows)
print(json.dumps(out, indent=1))
