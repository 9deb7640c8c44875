"""bench.py - the BASELINE.json workloads through HipExecutionEngine on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config q1|join|strkey] [--sf SF]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default = the metric's own configuration: TPC-H Q1 (the reference's benchmark query, README.md:141-158) on synthetic
lineitem sf=100.  A "step" is one pass of the hot path: ``DataFrame.collect()`` with the referenced columns already
resident in HBM (block b on rank b % N) -> result rows on the host of rank 0.  Strong scaling: the table is fixed and
split across the ranks.  Rank 0 prints ONE JSON line (DESIGN.md section 6 explains every field).

``--config join`` / ``--config strkey`` run BASELINE configs 4 and 5 (sf=10 by default) with the same JSON shape, their
own byte accounting and their own C oracle ports (oracle/q45_oracle.c).
"""

from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import tempfile
import time
from datetime import datetime
from pathlib import Path

os.environ.setdefault("TZ", "UTC")
time.tzset()
ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s)
CUTOFF = "1998-12-01"  # the reference's query (README.md:141-158): TPC-H's last ship date, every row passes
SELECTIVE_CUTOFF = "1998-09-02"  # SURVEY 8d's second cutoff: ~97 % of the rows pass (other_configs.q1_cutoff)
Q1_MOVED_BYTES_PER_ROW = 25  # the length bytes of the fixed-width l_returnflag are never read


def q1_frame(engine, table_path: str, cutoff: str = CUTOFF):
    from minispark_amd import workloads

    return workloads.q1(workloads.engine_api(engine), table_path, cutoff)


SCAN_KERNEL_SOURCES = ("hs_device.h", "hs_agg_kernel.h", "hs_capture.h", "hs_agg.hip", "hs_jit.hip")
# per bench config: the dominant kernel whose PMC traffic is quoted, the other kernels of a step worth a line, the sources
# whose hash stamps the measurement, the committed file (tools/pmc_traffic.py writes it from this command's --pmc passes)
TRAFFIC_KERNELS = {
    "q1": {"dominant": "k_agg_jit", "others": ("k_agg_finish",), "sources": SCAN_KERNEL_SOURCES,
           "file": "r04_pmc_hbm_traffic_q1_sf100.json"},
    "strkey": {"dominant": "k_agg_jit", "others": ("k_dict_combine", "k_agg_finish"), "sources": SCAN_KERNEL_SOURCES + ("hs_join.hip",),
               "file": "r04_pmc_hbm_traffic_strkey_sf10.json"},
    "join": {"dominant": "k_agg_shared_jit", "others": ("k_join8_", "k_agg_shared_fold_chunks", "k_agg_units_to_slab", "k_agg_finish"),
             "sources": SCAN_KERNEL_SOURCES + ("hs_join.hip",), "file": "r04_pmc_hbm_traffic_join_sf10.json"},
}


def kernel_sources_sha(sources=SCAN_KERNEL_SOURCES) -> str:
    """Identity of a scan kernel's code: PMC traffic figures under profiles/ are only quoted while it is unchanged.
    Covers what the kernel is compiled from and launched by - the embedded headers, the code generator, the launch
    geometry (csrc/hs_agg.hip), for the join and dictionary configs csrc/hs_join.hip, and the public header they all
    include; operators the step never launches (radix tier, stage engine) are not part of it."""
    h = hashlib.sha256()
    for p in [*(ROOT / "minispark_amd" / "csrc" / name for name in sources), ROOT / "include" / "hipspark.h"]:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()[:16]


def pmc_traffic(config: str) -> dict:
    """HBM bytes per launch of the config's dominant kernel from the committed rocprofv3 counter passes of THIS command
    (profiles/<file>; rocprofv3 cannot run inside the process) - refused when the kernels changed since."""
    spec = TRAFFIC_KERNELS[config]
    name = spec["file"]
    path = ROOT / "profiles" / name
    if not path.exists():
        return {"traffic": None, "traffic_source": f"profiles/{name} not collected yet"}
    t = json.loads(path.read_text())
    now = kernel_sources_sha(spec["sources"])
    if t.get("kernel_sources_sha") != now:
        return {"traffic": None, "traffic_source": f"profiles/{name} is stale: measured at kernel sources "
                                                   f"{t.get('kernel_sources_sha')}, now {now}"}
    return {"traffic": t["hbm_read_bytes_per_launch (FETCH_SIZE*1024*2)"] + t["hbm_write_bytes_per_launch"],
            "traffic_other_kernels": t.get("other_kernels") or None,
            "traffic_source": f"profiles/{name} (separate --pmc FETCH_SIZE / WRITE_SIZE passes at kernel sources "
                              f"{t['kernel_sources_sha']}, commit {t.get('commit', '?')})"}


def python_engine_baseline(scratch: Path, cutoff: str, n: int = 120_000) -> dict:
    """The row-at-a-time pure-Python restatement of the reference's PythonExecutionEngine (oracle/py_engine.py:
    BlockFile decode value by value, expression tree per row, dict aggregators) on a small sample, 1 core -
    the kind of number the reference's own Python engine produces (it cannot run on this box)."""
    import numpy as np

    from minispark_amd import synth
    from minispark_amd.io import BlockFile, StrCol
    from oracle import py_engine, q1_native

    cols = q1_native.gen(synth.SEED, 0, n)
    schema = [c for c in synth.LINEITEM_SCHEMA if c[0] != "l_orderkey"]
    raw = [cols["l_quantity"], cols["l_extendedprice"], cols["l_discount"], cols["l_tax"],
           StrCol(np.ones(n, dtype=np.uint8), cols["l_returnflag"]), cols["l_shipdate"]]
    path = scratch / "py_sample.bin"
    BlockFile(path, schema).write_raw(schema, raw)
    frame = q1_frame(object(), str(path), cutoff)
    t0 = time.perf_counter()
    rows = py_engine.run_query(frame.task)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "rows/s", "cores": 1, "rows": n, "groups": len(rows)}


def q1_cpu_baseline(table, sample_blocks: int, gpu_rows_for_sample, cutoff: str = CUTOFF):
    """The C port of the reference's algorithm (oracle/q1_oracle.c) timed on the host cores over the
    first ``sample_blocks`` blocks of the same table; also checks the GPU's rows for that sample."""
    from oracle import blockfile as bfio
    from oracle import q1_native
    from oracle.compare import assert_rows_match

    sizes = table.block_rows[:sample_blocks]
    n = sum(sizes)
    names = {1: "l_quantity", 2: "l_extendedprice", 3: "l_discount", 4: "l_tax", 5: "l_returnflag", 6: "l_shipdate"}
    cols = {name: table.stored_column(cid).data[:n].cpu().numpy() for cid, name in names.items()}
    cutoff_us = bfio.to_us(datetime.fromisoformat(cutoff))
    threads = q1_native.host_threads()
    q1_native.run(cols, sizes, cutoff_us, threads=threads)  # warm-up (page in, build)
    t0 = time.perf_counter()
    passes_mt = 0
    while time.perf_counter() - t0 < 6.0 or passes_mt < 2:
        want = q1_native.run(cols, sizes, cutoff_us, threads=threads)
        passes_mt += 1
    t_mt = (time.perf_counter() - t0) / passes_mt
    t0 = time.perf_counter()
    passes_1 = 0
    while time.perf_counter() - t0 < 6.0 or passes_1 < 1:
        want1 = q1_native.run(cols, sizes, cutoff_us, threads=1)
        passes_1 += 1
    t_1 = (time.perf_counter() - t0) / passes_1
    assert want == want1, "C oracle: threaded and scalar runs differ"
    flips = assert_rows_match(gpu_rows_for_sample, want, max_ulps=1)
    return {
        "value": n / t_mt, "unit": "rows/s", "cores": threads, "kind": "port",
        "sample": f"first {sample_blocks} blocks ({n} rows) of the same table; {passes_mt} passes on {threads} threads, "
                  f"{passes_1} on 1 thread",
        "value_1core": n / t_1,
        "gpu_matches_oracle_on_sample": True, "f32_ulp_flips_on_sample": flips,
    }


def q1_full_check(rows, total_rows: int, cutoff: str) -> dict:
    """The TIMED result against the C oracle over the WHOLE table: the oracle regenerates the synthetic table block
    by block with the generator's CPU twin (nothing of the 15.6 GB is resident) and runs the reference's algorithm -
    per-block fp64 partials, f32 / i32 shuffle write, block-ordered merge (oracle/q1_oracle.c q1_run_synth)."""
    from minispark_amd import constants, synth
    from oracle import blockfile as bfio
    from oracle import q1_native
    from oracle.compare import assert_rows_match

    threads = q1_native.host_threads()
    t0 = time.perf_counter()
    want = q1_native.run_synth(synth.SEED, total_rows, constants.ROWS_PER_BLOCK,
                               bfio.to_us(datetime.fromisoformat(cutoff)), threads=threads)
    dt = time.perf_counter() - t0
    try:
        flips = assert_rows_match(rows, want, max_ulps=1)
        ok, why = True, None
    except AssertionError as e:
        flips, ok, why = None, False, str(e)[:400]
    out = {"gpu_matches_oracle_full": ok, "f32_ulp_flips_full": flips, "rows_checked": total_rows,
           "oracle": f"oracle/q1_oracle.c q1_run_synth on {threads} host threads, {dt:.1f} s (generation included)"}
    if why:
        out["mismatch"] = why
    return out


def timed_steps(wl, engine, steps: int, warmup: int, dist):
    """W untimed + exactly K timed steps of one workload, bracketed by synchronise (+ barrier) on both sides; the MAX
    over the ranks.  -> (seconds for the K steps, dominant kernel ms per step, exchange ms per step, last rows).
    A workload may hand over several equal `frames` over copies of its table (data that fits the 256 MiB Infinity Cache:
    SURVEY 8d asks for >= 4 copies in rotation so that the cache does not serve the reads); step i runs frames[i % n]."""
    import torch

    frames = getattr(wl, "frames", None) or [wl.frame]
    engine.dev.time_scan_kernel(True)
    if dist is not None:
        engine.dev.time_exchange(True)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # setup, not warm-up: the first runs of a query compile its kernels (hiprtc), fill the plan / launch caches and
    # record the replayable launch sequence; whatever --warmup says, the timed steps are steady-state steps.  What the
    # set-up costs is reported, not hidden: `cold`
    lib = engine.dev._raw_lib
    jit0, dict0 = lib.hs_jit_compile_seconds(), getattr(engine.dev, "dict_encode_seconds", 0.0)
    setup_ms = []
    for frame in frames:
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            frame.collect()
            setup_ms.append((time.perf_counter() - t0) * 1e3)
    wl.cold = {"first_collect_ms": setup_ms[0], "second_collect_ms (recorded)": setup_ms[1],
               "third_collect_ms (first replay)": setup_ms[2],
               "jit_ms": (lib.hs_jit_compile_seconds() - jit0) * 1e3, "jit_disk_hits": int(lib.hs_jit_disk_hits()),
               "dict_code_ms": (getattr(engine.dev, "dict_encode_seconds", 0.0) - dict0) * 1e3}
    rows = None
    for i in range(warmup):
        rows = frames[i % len(frames)].collect()
    fence()
    kernel_ms, exchange_ms = [], []
    t0 = time.perf_counter()
    for i in range(steps):
        rows = frames[i % len(frames)].collect()
        kernel_ms.append(wl.dominant_kernel_ms())  # the step ended with the result on the host: events are complete
        if dist is not None:
            exchange_ms.append(wl.exchange_ms())
    fence()
    elapsed = time.perf_counter() - t0
    kernel_avg_ms = sum(kernel_ms) / len(kernel_ms)
    exchange_avg_ms = sum(exchange_ms) / len(exchange_ms) if exchange_ms else 0.0
    if dist is not None:
        t = torch.tensor([elapsed, kernel_avg_ms, exchange_avg_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_avg_ms, exchange_avg_ms = (float(v) for v in t.tolist())
    return elapsed, kernel_avg_ms, exchange_avg_ms, rows


def _other_line(wl, engine, steps: int, warmup: int) -> dict:
    elapsed, kernel_avg_ms, _, rows = timed_steps(wl, engine, steps, warmup, None)
    ms = elapsed / steps * 1e3
    extra = wl.extra_split_ms() if hasattr(wl, "extra_split_ms") else {}
    return {"metric": wl.metric, "value": wl.total_units / (elapsed / steps), "unit": "rows/s", "steps": steps,
            "ms_per_step": ms, "config": wl.config(rows), "roofline": wl.roofline(kernel_avg_ms),
            "time_split_ms": dict({"scan_partial": kernel_avg_ms, "exchange": 0.0,
                                   "final": ms - kernel_avg_ms - sum(extra.values())}, **extra),
            "cold": wl.cold, "full_check": wl.full_check(rows)}


def other_configs(headline, local_rank: int, scratch: Path, steps: int, warmup: int) -> dict:
    """What else the driver's line must carry, in the same process AFTER the headline's timed region (the driver runs the
    default command only; these lines must not exist as builder-run files alone):
      q1_cutoff  the headline's table under WHERE l_shipdate <= '1998-09-02' (~97 % of the rows: SURVEY 8d, the predicate
                 is not vacuous);
      q1_sf1     BASELINE config 2 (Q1 on sf=1): 156 MB - four copies of the table in rotation so that the Infinity Cache
                 does not serve the reads, and the one-table (cache-resident) figure beside it;
      join, strkey   BASELINE configs 4 and 5 (sf=10), each with its own engine and tables.
    Each: `steps` timed steps like the headline, the timed result checked against the C port over the whole tables."""
    from types import SimpleNamespace

    from minispark_amd.execution import HipExecutionEngine
    from tools.bench_configs import JoinWorkload, StrKeyWorkload

    out = {}
    engine = headline.engine
    a = headline.args
    wl = Q1Workload(engine, scratch, SimpleNamespace(sf=a.sf, cutoff=SELECTIVE_CUTOFF, sample_blocks=a.sample_blocks), 0, 1,
                    table=(headline.table_path, headline.table))
    out["q1_cutoff"] = _other_line(wl, engine, steps, warmup)
    wl = Q1Workload(engine, scratch, SimpleNamespace(sf=1.0, cutoff=CUTOFF, sample_blocks=a.sample_blocks), 0, 1, copies=4)
    line = _other_line(wl, engine, steps, warmup)
    wl.frames = wl.frames[:1]
    resident = timed_steps(wl, engine, steps, warmup, None)
    line["cache_resident"] = {"ms_per_step": resident[0] / steps * 1e3, "kernel_ms": resident[1],
                              "note": "the same query on ONE copy of the table (156 MB: served by the Infinity Cache)"}
    out["q1_sf1"] = line
    for name, cls in (("join", JoinWorkload), ("strkey", StrKeyWorkload)):
        engine = HipExecutionEngine(device=local_rank)
        try:
            wl = cls(engine, scratch / f"other_{name}", SimpleNamespace(sf=10.0, config=name), 0, 1)
            out[name] = _other_line(wl, engine, steps, warmup)
        finally:
            engine.__exit__(None, None, None)
            del engine
    return out


def second_process_cold(args) -> dict:
    """A second process on the same box (code objects of the first on disk: HIPSPARK_JIT_CACHE): builds the same table and
    times its FIRST collect().  A child started with subprocess - the parent keeps its tables; 2 processes on the GPU."""
    import subprocess

    cmd = [sys.executable, str(Path(__file__).resolve()), "--cold-child", "--sf", str(args.sf), "--cutoff", args.cutoff]
    try:
        run = subprocess.run(cmd, capture_output=True, text=True, timeout=300, check=False)
        lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
        if run.returncode != 0 or not lines:
            return {"error": f"rc {run.returncode}: {run.stderr[-300:]}"}
        return json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        return {"error": "timed out after 300 s"}


def cold_child(args) -> None:
    import torch

    from minispark_amd import constants
    from minispark_amd.execution import HipExecutionEngine

    scratch = Path(tempfile.mkdtemp(prefix="hipspark_cold_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None))
    constants.SHUFFLE_FOLDER = scratch / "shuffle"
    engine = HipExecutionEngine(device=0)
    wl = Q1Workload(engine, scratch, args, 0, 1)
    lib = engine.dev._raw_lib
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows = wl.frame.collect()
    first = (time.perf_counter() - t0) * 1e3
    print(json.dumps({"first_collect_ms": first, "jit_ms": lib.hs_jit_compile_seconds() * 1e3,
                      "jit_disk_hits": int(lib.hs_jit_disk_hits()), "groups": len(rows)}), flush=True)
    engine.__exit__(None, None, None)
    import shutil

    shutil.rmtree(scratch, ignore_errors=True)


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` as a plain command: start the N ranks as a CHILD `python -m torch.distributed.run`
    with the same arguments, let rank 0's JSON line through on stdout and hand the child's exit code back.  Called
    before this process has imported torch or made any HIP call (a process that has touched the GPU must not be
    replaced, and is not: the parent only waits)."""
    import socket
    import subprocess

    with socket.socket() as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
        "HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.run(cmd, env=env, check=False).returncode


def launch_probe(rank: int, world: int) -> None:
    """HIPSPARK_BENCH_LAUNCH_ONLY=1 (tests/test_distributed_cpu.py): the ranks rendezvous, agree on the world size
    and leave - proves the launch path of `--gpus N` on a box without N GPUs; no kernel, no engine."""
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(os.environ.get("HIPSPARK_DIST_BACKEND", "nccl"))
    t = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launched_ranks": int(t.item()), "world": world}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["q1", "join", "strkey"], default="q1")
    ap.add_argument("--sf", type=float, default=None, help="scale factor (default: 100 for q1, 10 for join / strkey)")
    ap.add_argument("--sample-blocks", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-check", action="store_true", help="skip the whole-table oracle check of the timed result")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="q1 on one GPU: do not run BASELINE configs 4 / 5 (sf=10) after the timed region")
    ap.add_argument("--cutoff", default=CUTOFF, help="q1: WHERE l_shipdate <= CUTOFF")
    ap.add_argument("--cold-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.sf is None:
        args.sf = 100.0 if args.config == "q1" else 10.0
    if args.cold_child:
        return cold_child(args)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))  # nothing has touched the GPU (torch is not even imported yet)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the two must agree")
    if os.environ.get("HIPSPARK_BENCH_LAUNCH_ONLY") == "1":
        return launch_probe(rank, world)
    # rehearsal knobs (one-GPU box): HIPSPARK_DIST_BACKEND=gloo HIPSPARK_FORCE_DEVICE=0 run N ranks on one GPU
    backend = os.environ.get("HIPSPARK_DIST_BACKEND", "nccl")
    if "HIPSPARK_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["HIPSPARK_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = os.environ.get("HIPSPARK_FORCE_DIST") == "1"  # exercise the collective path even at world 1
    if world > 1 or force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from minispark_amd import constants
    from minispark_amd.execution import HipExecutionEngine

    scratch = Path(tempfile.mkdtemp(prefix=f"hipspark_bench_r{rank}_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None))
    constants.SHUFFLE_FOLDER = scratch / "shuffle"
    if "HIPSPARK_JIT_CACHE" not in os.environ:
        # a code-object cache of this run's own: `cold.jit_ms` is then a real hiprtc compile, not a hit on what an earlier
        # process of the box left behind; the second process of `cold` (and the other configs' engines) find it warm
        os.environ["HIPSPARK_JIT_CACHE"] = str(scratch / "jit-cache")
    engine = HipExecutionEngine(device=local_rank)
    if dist is not None:
        engine.enable_distributed(dist)
    if args.config == "q1":
        wl = Q1Workload(engine, scratch, args, rank, world)
    else:
        from tools.bench_configs import JoinWorkload, StrKeyWorkload  # noqa: PLC0415

        wl = (JoinWorkload if args.config == "join" else StrKeyWorkload)(engine, scratch, args, rank, world)
    elapsed, kernel_avg_ms, exchange_avg_ms, rows = timed_steps(wl, engine, args.steps, args.warmup, dist)
    extra_split = wl.extra_split_ms() if hasattr(wl, "extra_split_ms") else {}
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": wl.metric, "value": wl.total_units / (elapsed / args.steps), "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": dict(wl.config(rows), placement="block b on rank b % n_gpus",
                           exchange="none" if dist is None else wl.exchange_text(backend)),
            "roofline": wl.roofline(kernel_avg_ms),
            # where a step goes (SURVEY.md 8d "Scaling report"): scan + partial aggregate (the dominant kernel with the
            # per-unit combine fused into its epilogue; HIP events), the exchange collective (events around it on the
            # launch stream; 0 without ranks), and the final merge + projection + hand-over + host time (the rest)
            "time_split_ms": dict({"scan_partial": kernel_avg_ms, "exchange": exchange_avg_ms,
                                   "final": ms_per_step - kernel_avg_ms - exchange_avg_ms - sum(extra_split.values())},
                                  **extra_split),
            "whole_step_GBps_per_gpu": wl.algorithmic_bytes_per_launch() / (elapsed / args.steps) / 1e9,
        }
        out["cold"] = dict(wl.cold)
        if not args.no_full_check:
            out["full_check"] = wl.full_check(rows)
        if world == 1 and dist is None and not args.no_cpu_baseline:
            out["cpu_baseline"] = wl.cpu_baseline()
        if world == 1 and dist is None and args.config == "q1" and not args.no_other_configs and not args.no_full_check:
            out["cold"]["second_process_first_collect_ms"] = second_process_cold(args)
            out["other_configs"] = other_configs(wl, local_rank, scratch, args.steps, args.warmup)
        print(json.dumps(out), flush=True)
    engine.__exit__(None, None, None)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    import shutil

    shutil.rmtree(scratch, ignore_errors=True)


class Q1Workload:
    """BASELINE configs 2 / 3: Q1 on synthetic lineitem generated in HBM (counter-based, block b on rank b % world)."""

    metric = "TPC-H Q1 lineitem rows/sec"

    def __init__(self, engine, scratch: Path, args, rank: int, world: int, table=None, copies: int = 1) -> None:
        from minispark_amd import synth

        self.engine, self.scratch, self.args, self.world = engine, scratch, args, world
        self.total_units = synth.lineitem_rows(args.sf)
        if table is not None:  # another query over a table that is already resident
            self.table_path, self.table = table
        else:
            self.table_path = scratch / f"lineitem_sf{args.sf:g}.bin"
            self.table = synth.make_lineitem(engine.dev, self.table_path, self.total_units, rank=rank, world=world)
            engine.attach_device_table(self.table_path, self.table)
        self.frame = q1_frame(engine, str(self.table_path), args.cutoff)
        self.frames = [self.frame]
        for k in range(1, copies):  # equal tables at other addresses (timed_steps rotates among them)
            path = scratch / f"lineitem_sf{args.sf:g}_copy{k}.bin"
            engine.attach_device_table(path, synth.make_lineitem(engine.dev, path, self.total_units, rank=rank, world=world))
            self.frames.append(q1_frame(engine, str(path), args.cutoff))

    def dominant_kernel_ms(self) -> float:
        return self.engine.dev.scan_kernel_ms()

    def exchange_ms(self) -> float:
        return self.engine.dev.exchange_ms()

    def exchange_text(self, backend: str) -> str:
        if os.environ.get("HIPSPARK_P2P_SLABS") == "1":
            return (f"partial-row slabs stored into every peer's hipIpc-mapped buffer + device-side flags, no collective "
                    f"(prototype; process group: {backend})")
        return f"one all_gather of partial-row slabs per query ({backend})"

    def algorithmic_bytes_per_launch(self) -> float:
        from minispark_amd import synth

        return synth.Q1_BYTES_PER_ROW * self.table.nrows

    def config(self, rows) -> dict:
        from minispark_amd import constants, synth

        a = self.args
        return {
            "workload": f"TPC-H Q1 variant (8 aggregates, WHERE l_shipdate <= '{a.cutoff}', GROUP BY l_returnflag) "
                        f"on synthetic lineitem sf={a.sf:g}",
            "rows": self.total_units, "blocks": len(synth.block_sizes(self.total_units)),
            "rows_per_block": constants.ROWS_PER_BLOCK, "bytes_per_row": synth.Q1_BYTES_PER_ROW,
            "groups": len(rows or []), "rows_passing_the_predicate": sum(int(r["count_order"]) for r in rows or []),
            "table_copies_in_rotation": len(self.frames),
        }

    def roofline(self, kernel_avg_ms: float) -> dict:
        from minispark_amd import synth

        local_rows = self.table.nrows
        achieved = synth.Q1_BYTES_PER_ROW * local_rows / (kernel_avg_ms * 1e-3) / 1e9
        out = {
            "bound": "hbm", "kernel": "k_agg_jit (k_agg_main body; unit combine fused into its epilogue)",
            "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None, "kernel_ms": kernel_avg_ms, "rows_per_launch": local_rows,
            # `achieved` prices the ALGORITHMIC 26 B/row; the kernel moves 25 (fixed-width key: length
            # bytes are not read), so the rate at which bytes really leave HBM is the lower figure below
            "algorithmic_bytes_per_row": synth.Q1_BYTES_PER_ROW, "moved_bytes_per_row": Q1_MOVED_BYTES_PER_ROW,
            "moved_GBps": Q1_MOVED_BYTES_PER_ROW * local_rows / (kernel_avg_ms * 1e-3) / 1e9,
            "launch": self.engine.dev.last_scan,
        }
        if self.world == 1 and self.total_units == 600_037_902 and self.args.cutoff == CUTOFF:
            out.update(pmc_traffic("q1"))
            if out["traffic"]:
                out["hbm_frac"] = out["traffic"] / (kernel_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
        return out

    def full_check(self, rows) -> dict:
        return q1_full_check(rows, self.total_units, self.args.cutoff)

    def cpu_baseline(self) -> dict:
        from minispark_amd import synth

        a = self.args
        blocks = min(a.sample_blocks, len(self.table.block_rows))
        sample = synth.make_lineitem(self.engine.dev, self.scratch / "sample.bin", sum(self.table.block_rows[:blocks]))
        self.engine.attach_device_table(self.scratch / "sample.bin", sample)
        gpu_sample_rows = q1_frame(self.engine, str(self.scratch / "sample.bin"), a.cutoff).collect()
        out = q1_cpu_baseline(self.table, blocks, gpu_sample_rows, a.cutoff)
        out["python_engine_port"] = python_engine_baseline(self.scratch, a.cutoff)
        return out


if __name__ == "__main__":
    main()
