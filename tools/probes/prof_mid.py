"""Host profile of the 2 526-group GROUP BY (60 M rows): python tools/probes/prof_mid.py"""
import os, sys, tempfile, time, cProfile, pstats
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
table = synth.make_lineitem(engine.dev, path, 60_000_000)
engine.attach_device_table(path, table)
q = DataFrame(engine).table(str(path)).group_by(Col("l_shipdate")).agg(F.sum(Col("l_extendedprice")).alias("s"), F.count())
for i in range(4):
    q.collect()
N = 20
t0 = time.perf_counter()
for i in range(N):
    q.collect()
print(f"{(time.perf_counter() - t0) / N * 1e3:.3f} ms/query")
pr = cProfile.Profile(); pr.enable()
for i in range(N):
    q.collect()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative")
import io
buf = io.StringIO(); st.stream = buf; st.print_stats(38); print(buf.getvalue()[:6000])
engine.__exit__(None, None, None)
