"""Timing of the generic row exchange (HipExecutionEngine._exchange_by_key: partition ids -> stable sort by destination
-> pack -> size matrix -> all_to_all_single -> unpack) on N ranks:

    HIPSPARK_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 \
        --master-port P tools/bench_exchange.py [rows per rank] [--legacy /tmp/round2_execution.py]   (git show 9993950:minispark_amd/execution.py > /tmp/round2_execution.py: the before/after of profiles/r03_exchange_pack_3ranks_gloo.txt)

An e2e_join_select-shaped batch per rank: INTEGER key, FLOAT value, a variable-length STRING.  --legacy FILE binds the
_exchange_rows of another version of minispark_amd/execution.py (round 2's Python-side packing) for a before / after on
the same box.  Under gloo on one GPU the collective itself goes through the host: the figure that compares is
`pack_unpack_ms` = the step minus the two collectives."""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("TZ", "UTC")
time.tzset()
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main() -> None:
    import numpy as np
    import torch
    import torch.distributed as dist

    rows = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10_000_000
    legacy = sys.argv[sys.argv.index("--legacy") + 1] if "--legacy" in sys.argv else None
    torch.cuda.set_device(int(os.environ.get("HIPSPARK_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    dist.init_process_group(os.environ.get("HIPSPARK_DIST_BACKEND", "nccl"))
    rank, world = dist.get_rank(), dist.get_world_size()

    from minispark_amd import distributed as D
    from minispark_amd import hipspark as hs
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.device import DBatch, DCol
    from minispark_amd.execution import HipExecutionEngine

    engine = HipExecutionEngine(device=torch.cuda.current_device())
    engine.enable_distributed(dist)
    if legacy:
        spec = importlib.util.spec_from_file_location("minispark_amd._legacy_execution", legacy)
        mod = importlib.util.module_from_spec(spec)
        mod.__package__ = "minispark_amd"
        spec.loader.exec_module(mod)
        engine._exchange_rows = mod.HipExecutionEngine._exchange_rows.__get__(engine)
    dev = engine.dev
    rng = np.random.default_rng(100 + rank)
    key = dev.fixed_col(hs.I32, rng.integers(0, 1 << 30, rows, dtype=np.int64).astype(np.int32))
    val = dev.fixed_col(hs.F32, rng.random(rows, dtype=np.float32))
    lens = rng.integers(0, 13, rows).astype(np.uint8)
    data = rng.integers(97, 123, int(lens.sum(dtype=np.int64)), dtype=np.uint8)
    text = dev.string_col(dev.to_device(lens, torch.uint8), dev.to_device(data, torch.uint8), rows)
    batch = DBatch([("k", T.INTEGER), ("v", T.FLOAT), ("s", T.STRING)], [key, val, text], rows, [0, rows])
    # time spent inside the two collectives (host-staged under gloo), so that pack + unpack can be told apart
    spent = {"t": 0.0}

    def timed(fn):
        def inner(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(*a, **k)
            torch.cuda.synchronize()
            spent["t"] += time.perf_counter() - t0
            return out
        return inner

    for name in ("all_to_all_rows", "exchange_size_matrix", "exchange_counts"):
        setattr(D, name, timed(getattr(D, name)))
    steps, coll = [], []
    got = None
    for _ in range(6):
        dist.barrier()
        torch.cuda.synchronize()
        spent["t"] = 0.0
        t0 = time.perf_counter()
        got, part = engine._exchange_by_key(batch, 0)
        torch.cuda.synchronize()
        steps.append((time.perf_counter() - t0) * 1e3)
        coll.append(spent["t"] * 1e3)
    check = [int(got.nrows), int(got.cols[0].data[: got.nrows].to(torch.int64).sum().item()),
             int(got.cols[2].lens[: got.nrows].to(torch.int64).sum().item())]
    mine = torch.tensor([min(steps[1:]), min(s - c for s, c in zip(steps[1:], coll[1:]))] + [float(v) for v in check], dtype=torch.float64)
    everyone = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    if rank == 0:
        print(json.dumps({"what": "row exchange by key", "packing": "round 2 (Python slices + torch.cat)" if legacy else "device (hs_copy_segments)",
                          "world": world, "rows_per_rank": rows, "backend": dist.get_backend(),
                          "step_ms_max_over_ranks": max(float(t[0]) for t in everyone),
                          "pack_unpack_ms_max_over_ranks": max(float(t[1]) for t in everyone),
                          "received_rows": [int(t[2]) for t in everyone], "key_sums": [int(t[3]) for t in everyone],
                          "string_bytes": [int(t[4]) for t in everyone]}))
    engine.__exit__(None, None, None)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
