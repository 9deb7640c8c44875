"""Host-side profile of one Q1 step: python tools/profile_step.py [sf] (small sf: kernel time negligible).
Prints the step time and the cumulative microseconds per call of the host functions on the path."""
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
for _ in range(5): frame.collect()
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N): frame.collect()
print(f"sf={sf:g}: {(time.perf_counter() - t0) / N * 1e6:.1f} us/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(N): frame.collect()
pr.disable()
st = pstats.Stats(pr)
rows = []
for (fn, line, name), (cc, nc, tt, ct, _) in st.stats.items():
    rows.append((ct / N * 1e6, tt / N * 1e6, nc / N, f"{Path(fn).name}:{line}({name})"))
rows.sort(reverse=True)
print(f"{'cum us/step':>12s} {'own us/step':>12s} {'calls/step':>10s}  function (under cProfile: inflated ~2x)")
for ct, tt, nc, name in rows[:40]:
    print(f"{ct:12.1f} {tt:12.1f} {nc:10.1f}  {name}")
