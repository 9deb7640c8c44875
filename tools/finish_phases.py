"""Phase times inside k_agg_finish (wall_clock64 stamps, 100 MHz): HIPSPARK_FINISH_STAMPS=1 python tools/finish_phases.py [sf]"""
import os, sys, tempfile, time
from pathlib import Path
os.environ["HIPSPARK_FINISH_STAMPS"] = "1"
os.environ.setdefault("TZ", "UTC"); time.tzset()
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import q1_frame
from minispark_amd import constants, synth
from minispark_amd.execution import HipExecutionEngine
scratch = Path(tempfile.mkdtemp(prefix="hs_ph_", dir="/dev/shm"))
constants.SHUFFLE_FOLDER = scratch / "shuffle"
sf = float(sys.argv[1]) if len(sys.argv) > 1 else 12.5
engine = HipExecutionEngine(0)
path = scratch / "li.bin"
table = synth.make_lineitem(engine.dev, path, synth.lineitem_rows(sf))
engine.attach_device_table(path, table)
frame = q1_frame(engine, str(path))
for _ in range(8): frame.collect()
torch.cuda.synchronize()
names = ["args+slabs staged", "core: rows->slots", "core: scans", "core: visiting seq", "core: ranks", "core: scatter",
         "core: fold", "key scratch", "pass-through cols", "projection", "header+image copy", "system fence"]
for prep in engine.dev._finish_prepared.values():
    cap = engine.dev.last_merge_cap
    n_fold = prep["fin"].n_fold
    off = (cap * 8 * (n_fold + 2) + 64 + 7) & ~7
    st = prep["scratch"][off: off + 13 * 8].view(torch.int64).cpu().tolist()
    for i, name in enumerate(names):
        print(f"{name:24s} {(st[i + 1] - st[i]) / 100:7.2f} us")
    print(f"{'total inside kernel':24s} {(st[12] - st[0]) / 100:7.2f} us")
