"""Timing of a handful of query shapes at sf=10 (60 M rows) through the engine - a hunt for slow paths."""
import os, sys, tempfile, time
from pathlib import Path
os.environ.setdefault("TZ", "UTC"); time.tzset()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from minispark_amd import constants, synth
from minispark_amd.dataframe import DataFrame
from minispark_amd.execution import HipExecutionEngine
from minispark_amd.sql import Col, Functions as F, Lit
scratch = Path(tempfile.mkdtemp(prefix="hs_sh_", dir="/dev/shm"))
constants.SHUFFLE_FOLDER = scratch / "shuffle"
engine = HipExecutionEngine(0)
path = scratch / "li.bin"
table = synth.make_lineitem(engine.dev, path, 59_986_052, with_orderkey=True, with_shipmode=True)
engine.attach_device_table(path, table)
T = lambda: DataFrame(engine).table(str(path))
shapes = {
    "shipmode (7 string groups) sum": T().group_by(Col("l_shipmode")).agg(F.sum(Col("l_quantity")).alias("s")),
    "returnflag min/max/count": T().group_by(Col("l_returnflag")).agg(F.min(Col("l_extendedprice")).alias("lo"), F.max(Col("l_discount")).alias("hi"), F.count()),
    "computed key orderkey % 97": T().select((Col("l_orderkey") % 97).alias("m"), Col("l_quantity")).group_by(Col("m")).agg(F.sum(Col("l_quantity")).alias("s"), F.count()),
    "filter qty>25, group by tax (9 float groups)": T().filter(Col("l_quantity") > 25.0).group_by(Col("l_tax")).agg(F.avg(Col("l_extendedprice")).alias("a"), F.count()),
    "group by discount x returnflag? (computed string)": T().select((Col("l_returnflag") + "-" + Col("l_shipmode")).alias("k"), Col("l_quantity")).group_by(Col("k")).agg(F.sum(Col("l_quantity")).alias("s")),
    "like filter + count by returnflag": T().filter(Col("l_shipmode").like("%AIR%")).group_by(Col("l_returnflag")).agg(F.count()),
}
for name, q in shapes.items():
    ts = []
    r0 = engine.replays
    for i in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = q.collect(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name:55s} first {ts[0]:8.1f} ms  steady {min(ts[2:]):7.3f} ms  rows {len(out):5d}  replays {engine.replays - r0}  tier {engine.dev.last_scan.get('tier') if engine.dev.last_scan else None}", flush=True)
engine.__exit__(None, None, None)
