"""Host-side profile of a mid-cardinality GROUP BY (shared-dictionary tier + final merge): where the time of a
steady-state collect() goes.  usage: profile_midcard.py [rows]"""
import cProfile, os, pstats, sys, tempfile, time
from pathlib import Path
os.environ.setdefault("TZ", "UTC"); time.tzset()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from minispark_amd import constants, synth
from minispark_amd.dataframe import DataFrame
from minispark_amd.execution import HipExecutionEngine
from minispark_amd.sql import Col, Functions as F
scratch = Path(tempfile.mkdtemp(prefix="hs_mc_", dir="/dev/shm"))
constants.SHUFFLE_FOLDER = scratch / "shuffle"
engine = HipExecutionEngine(0)
path = scratch / "li.bin"
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 60_000_000
table = synth.make_lineitem(engine.dev, path, rows)
engine.attach_device_table(path, table)
for name, key in [("l_shipdate (2526 groups)", Col("l_shipdate")), ("l_quantity (50 groups)", Col("l_quantity"))]:
    q = DataFrame(engine).table(str(path)).group_by(key).agg(F.sum(Col("l_extendedprice")).alias("s"), F.count())
    for i in range(5):
        q.collect()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10):
        out = q.collect()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {dt*1e3:.3f} ms/collect groups={len(out)} replays={engine.replays}", flush=True)
    prof = cProfile.Profile(); prof.enable()
    for i in range(10):
        q.collect()
    prof.disable()
    pstats.Stats(prof).sort_stats("cumulative").print_stats(32)
engine.__exit__(None, None, None)
