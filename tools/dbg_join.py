import os, sys, time, tempfile
from pathlib import Path
os.environ.setdefault("TZ","UTC"); time.tzset()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from minispark_amd import constants, hipspark as hs, synth
from minispark_amd.execution import HipExecutionEngine
scratch = Path(tempfile.mkdtemp(prefix="hs_dbg_", dir="/dev/shm"))
constants.SHUFFLE_FOLDER = scratch / "shuffle"
engine = HipExecutionEngine(0)
dev = engine.dev
for n in (1_500_304, 14_996_513):
    t = synth.make_orders(dev, scratch / f"o{n}.bin", n)
    col = t.columns[1]
    for it in range(6):
        t0 = time.perf_counter()
        coded = dev.dict_encode(col)
        torch.cuda.synchronize()
        print(n, it, None if coded is None else coded.dict, f"{(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
