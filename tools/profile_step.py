"""Host-side profile of one Q1 step (small table so the kernel time is negligible)."""
import cProfile, os, pstats, sys, tempfile, time
from pathlib import Path
os.environ.setdefault("TZ", "UTC"); time.tzset()
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import q1_frame
from minispark_amd import constants, synth
from minispark_amd.execution import HipExecutionEngine
scratch = Path(tempfile.mkdtemp(prefix="hs_prof_", dir="/dev/shm"))
constants.SHUFFLE_FOLDER = scratch / "shuffle"
sf = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
engine = HipExecutionEngine(0)
path = scratch / "li.bin"
table = synth.make_lineitem(engine.dev, path, synth.lineitem_rows(sf))
engine.attach_device_table(path, table)
frame = q1_frame(engine, str(path))
for _ in range(3): frame.collect()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): frame.collect()
print("ms/step", (time.perf_counter() - t0) / 20 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): frame.collect()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
