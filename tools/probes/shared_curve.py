"""Shared-dictionary tier throughput against the number of groups (computed key orderkey % g), 60 M rows."""
import os, sys, tempfile, time
from pathlib import Path
os.environ.setdefault("TZ", "UTC"); time.tzset()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from minispark_amd import constants, synth
from minispark_amd.dataframe import DataFrame
from minispark_amd.execution import HipExecutionEngine
from minispark_amd.sql import Col, Functions as F
scratch = Path(tempfile.mkdtemp(prefix="hs_sc_", dir="/dev/shm"))
constants.SHUFFLE_FOLDER = scratch / "shuffle"
engine = HipExecutionEngine(0)
path = scratch / "li.bin"
table = synth.make_lineitem(engine.dev, path, 59_986_052, with_orderkey=True)
engine.attach_device_table(path, table)
engine.dev.time_scan_kernel(True)
for g in [int(a) for a in sys.argv[1:]] or [8, 17, 32, 64, 97, 256, 1000, 3000]:
    q = (DataFrame(engine).table(str(path)).select((Col("l_orderkey") % g).alias("m"), Col("l_quantity"), Col("l_extendedprice"))
         .group_by(Col("m")).agg(F.sum(Col("l_quantity")).alias("s"), F.sum(Col("l_extendedprice")).alias("p"), F.count()))
    ts = []
    for i in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = q.collect(); ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    print(f"groups {g:5d}: steady {min(ts[3:]):7.3f} ms  scan kernel {engine.dev.scan_kernel_ms():6.3f} ms  rows {len(out)}  tier {engine.dev.last_scan.get('tier')} wg {engine.dev.last_scan.get('wg_threads')} cap {engine.dev.last_scan.get('group_cap')}", flush=True)
engine.__exit__(None, None, None)
