"""HBM tier of GROUP BY (any cardinality; bit-exact sequential fold) on a multi-block table:
    python tools/bench_hbm_tier.py [lineitem_rows] [groups_per_block ...]
GROUP BY a key with ~g distinct values per file block (beyond the 4096-slot LDS tier), SUM + COUNT, timing the partial
aggregate (Device.aggregate_partial_global: one pass over all blocks since round 2), the whole query with its rows as
dicts (DataFrame.collect) and with its result column-wise (DataFrame.collect_columns, round 3)."""
import os, sys, tempfile, time
from pathlib import Path
os.environ.setdefault("TZ", "UTC"); time.tzset()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from minispark_amd import constants, hipspark as hs, synth
from minispark_amd.constants import ColumnType as T
from minispark_amd.dataframe import DataFrame
from minispark_amd.device import DCol
from minispark_amd.execution import HipExecutionEngine
from minispark_amd.io import BlockFile
from minispark_amd.sql import Col, Functions as F
from minispark_amd.table import DeviceTable

rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 59_986_052
gs = [int(float(g)) for g in sys.argv[2:]] or [20_000, 500_000]
scratch = Path(tempfile.mkdtemp(prefix="hs_hbm_", dir="/dev/shm"))
constants.SHUFFLE_FOLDER = scratch / "shuffle"
engine = HipExecutionEngine(0)
dev = engine.dev
li = synth.make_lineitem(dev, scratch / "li.bin", rows, with_orderkey=True)
for g in gs:
    # key = orderkey-derived value with ~g distinct values inside every block
    schema = [("k", T.INTEGER), ("v", T.FLOAT)]
    path = scratch / f"t{g}.bin"
    BlockFile(path, schema).write_rows([])
    k = (li.columns[0].data[:rows].to(torch.int64) * 2654435761 % g).to(torch.int32)
    kk = dev.empty(rows, torch.int32); kk.copy_(k)
    t = DeviceTable(path, schema, synth.block_sizes(rows), {0: DCol(hs.I32, kk, rows), 1: li.columns[2]}, ())
    engine.attach_device_table(path, t)
    inner = dev.aggregate_partial_global
    ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    def timed(*a, **kw):
        ev[0].record(); out = inner(*a, **kw); ev[1].record(); return out
    dev.aggregate_partial_global = timed
    q = DataFrame(engine).table(str(path)).group_by(Col("k")).agg(F.sum(Col("v")).alias("s"), F.count())
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = q.collect(); dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f"{g} groups/block x {len(t.block_rows)} blocks: run {i}: query {dt*1e3:8.2f} ms  partial aggregate {ev[0].elapsed_time(ev[1]):8.2f} ms "
              f"= {rows/ev[0].elapsed_time(ev[1])/1e6:.2f} G rows/s  result groups {len(out)}", flush=True)
    # the same query with the result handed over column-wise (no row dicts): what the engine itself costs end to end
    for i in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); cols = q.collect_columns(); dt = time.perf_counter() - t0
        print(f"{g} groups/block: columns run {i}: query {dt*1e3:8.2f} ms  ({len(cols['k'])} groups, numpy columns {sorted(cols)})", flush=True)
    import numpy as np
    order = np.argsort(cols["k"], kind="stable")
    by_key = sorted(out, key=lambda r: r["k"])
    assert [int(v) for v in cols["k"][order]] == [r["k"] for r in by_key]
    assert [float(v) for v in cols["s"][order]] == [r["s"] for r in by_key] and [int(v) for v in cols["count"][order]] == [r["count"] for r in by_key]
    dev.aggregate_partial_global = inner
engine.__exit__(None, None, None)
