"""Host-side cost of one replayed DataFrame.collect() of Q1 (cProfile over many steps on a small table: the GPU work is
short, what is left is the per-query host path): python tools/probes/profile_collect.py [sf] [steps]"""
import cProfile, os, pstats, sys, tempfile, time
from pathlib import Path
os.environ.setdefault("TZ", "UTC"); time.tzset()
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from bench import q1_frame
from minispark_amd import constants, synth
from minispark_amd.execution import HipExecutionEngine

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
scratch = Path(tempfile.mkdtemp(prefix="hs_pc_", dir="/dev/shm"))
constants.SHUFFLE_FOLDER = scratch / "shuffle"
engine = HipExecutionEngine(0)
path = scratch / "li.bin"
table = synth.make_lineitem(engine.dev, path, synth.lineitem_rows(sf))
engine.attach_device_table(path, table)
frame = q1_frame(engine, str(path))
for _ in range(20):
    frame.collect()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    frame.collect()
dt = (time.perf_counter() - t0) / steps
print(f"collect(): {dt * 1e6:.1f} us per step at sf={sf} (replays {engine.replays})")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    frame.collect()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
