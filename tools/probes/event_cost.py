"""What does timing the scan kernel cost a step?  Steps per second of Q1 at sf=SF with and without the event pair on the scan
launch (bench.py always has it on: the roofline line needs the kernel's duration).  usage: event_cost.py [sf=1]"""
import os, sys, tempfile, time
from pathlib import Path
os.environ.setdefault("TZ", "UTC"); time.tzset()
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from bench import q1_frame
from minispark_amd import constants, synth
from minispark_amd.execution import HipExecutionEngine
scratch = Path(tempfile.mkdtemp(prefix="hs_ev_", dir="/dev/shm"))
constants.SHUFFLE_FOLDER = scratch / "shuffle"
sf = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
for timed in (False, True, False, True):
    engine = HipExecutionEngine(0)
    path = scratch / f"li{int(timed)}.bin"
    table = synth.make_lineitem(engine.dev, path, synth.lineitem_rows(sf))
    engine.attach_device_table(path, table)
    if timed:
        engine.dev.time_scan_kernel(True)
    frame = q1_frame(engine, str(path))
    for _ in range(10): frame.collect()
    torch.cuda.synchronize()
    N = 300
    t0 = time.perf_counter()
    for _ in range(N): frame.collect()
    torch.cuda.synchronize()
    print(f"sf={sf:g} events {'on ' if timed else 'off'}: {(time.perf_counter() - t0) / N * 1e6:.1f} us/step", flush=True)
    engine.__exit__(None, None, None)
    del engine
