import os, sys, time, tempfile
from pathlib import Path
from types import SimpleNamespace
os.environ.setdefault("TZ","UTC"); time.tzset()
os.environ["HIPSPARK_REPLAY"] = "0"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from minispark_amd import constants
from minispark_amd.execution import HipExecutionEngine
from tools.bench_configs import JoinWorkload
scratch = Path(tempfile.mkdtemp(prefix="hs_dbg_", dir="/dev/shm"))
constants.SHUFFLE_FOLDER = scratch / "shuffle"
engine = HipExecutionEngine(0)
wl = JoinWorkload(engine, scratch, SimpleNamespace(sf=10.0, config="join"), 0, 1)
engine.dev.time_scan_kernel(True)
dev = engine.dev
pairs = {}
def wrap(obj, name):
    inner = getattr(obj, name)
    ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    ev[0].record(); ev[1].record()
    pairs[name] = ev
    def timed(*a, **k):
        ev[0].record(); out = inner(*a, **k); ev[1].record(); return out
    setattr(obj, name, timed)
for n in ("aggregate_partial", "aggregate_merge", "aggregate_merge_global", "download_batch", "quantise_cols", "gather_col"):
    wrap(dev, n)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rows = wl.frame.collect()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"run {i}: {1e3*(t1-t0):.2f} ms; join op {dev.join_ms():.3f}; scan kernel {dev.scan_kernel_ms():.3f}; " +
          f"{dev.last_scan} " + "; ".join(f"{n} {ev[0].elapsed_time(ev[1]):.3f}" for n, ev in pairs.items()), flush=True)
