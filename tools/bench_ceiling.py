"""How far is the Q1 scan kernel from what the same loads can reach?  Times, on the bench table (sf argument), the
scan kernel of (a) Q1 and (b) a query that reads the SAME seven columns (26 B/row) but does almost nothing with them
(one accumulator, no arithmetic to speak of): (b) is the load path's ceiling for this access pattern.
usage: bench_ceiling.py [sf]"""
import os, sys, tempfile, time
from pathlib import Path
os.environ.setdefault("TZ", "UTC"); time.tzset()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from minispark_amd import constants, synth
from minispark_amd.dataframe import DataFrame
from minispark_amd.execution import HipExecutionEngine
from minispark_amd.sql import Col, Functions as F, Lit
from bench import q1_frame

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
scratch = Path(tempfile.mkdtemp(prefix="hs_ceil_", dir="/dev/shm"))
constants.SHUFFLE_FOLDER = scratch / "shuffle"
engine = HipExecutionEngine(0)
rows = synth.lineitem_rows(sf)
path = scratch / "li.bin"
engine.attach_device_table(path, synth.make_lineitem(engine.dev, path, rows))
engine.dev.time_scan_kernel(True)
light = (DataFrame(engine).table(str(path)).filter(Col("l_shipdate") <= Lit("1998-12-01"))
         .group_by(Col("l_returnflag"))
         .agg(F.sum(Col("l_quantity") + Col("l_extendedprice") + Col("l_discount") + Col("l_tax")).alias("s")))
for name, frame in [("Q1", q1_frame(engine, str(path))), ("same columns, one sum", light)]:
    for _ in range(4):
        frame.collect()
    ms = []
    for _ in range(10):
        frame.collect()
        ms.append(engine.dev.scan_kernel_ms())
    avg = sum(ms) / len(ms)
    print(f"{name:24s} scan kernel {avg:.3f} ms  -> {26 * rows / avg / 1e6:.0f} GB/s ({26 * rows / avg / 1e6 / 80:.1f} % of 8 TB/s)  {engine.dev.last_scan}", flush=True)
engine.__exit__(None, None, None)
