"""Where does a launch of the Q1 scan kernel spend its time?  Every workgroup stamps wall_clock64() (100 MHz) at its phase
boundaries (HIPSPARK_SCAN_STAMPS=1, hs_agg_debug_scan_stamps); this prints when workgroups start, stream, finish.

    python tools/scan_stamps.py [sf=12.5]
"""
import os, sys, tempfile, time
from pathlib import Path
os.environ["HIPSPARK_SCAN_STAMPS"] = "1"
os.environ.setdefault("TZ", "UTC"); time.tzset()
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from bench import q1_frame
from minispark_amd import constants, synth
from minispark_amd.execution import HipExecutionEngine

scratch = Path(tempfile.mkdtemp(prefix="hs_st_", dir="/dev/shm"))
constants.SHUFFLE_FOLDER = scratch / "shuffle"
sf = float(sys.argv[1]) if len(sys.argv) > 1 else 12.5
engine = HipExecutionEngine(0)
path = scratch / "li.bin"
table = synth.make_lineitem(engine.dev, path, synth.lineitem_rows(sf))
engine.attach_device_table(path, table)
engine.dev.time_scan_kernel(True)
frame = q1_frame(engine, str(path))
for _ in range(8):
    frame.collect()
torch.cuda.synchronize()
lib = engine.dev._raw_lib
buf = np.zeros((1 << 16, 16), dtype=np.int64)
n = int(lib.hs_agg_debug_scan_stamps(buf.ctypes.data, buf.shape[0]))
st = buf[:n]
print(f"sf={sf:g}: {n} workgroups, scan kernel (events) {engine.dev.scan_kernel_ms() * 1e3:.1f} us, launch {engine.dev.last_scan}")
t = st[:, :7].astype(np.float64) / 100.0  # us
t0 = t[:, 0].min()
t -= t0
end = t[:, 6].max()
print(f"first entry -> last exit: {end:.1f} us")


def q(v):
    return " ".join(f"{np.percentile(v, p):7.1f}" for p in (0, 10, 50, 90, 100))


print("                                       min     p10     p50     p90     max   (us)")
print(f"entry (since first entry)         {q(t[:, 0])}")
print(f"entry -> early-exit check done    {q(t[:, 1] - t[:, 0])}")
print(f"check -> tables initialised       {q(t[:, 2] - t[:, 1])}")
print(f"streaming loop                    {q(t[:, 3] - t[:, 2])}")
print(f"loop end (since first entry)      {q(t[:, 3])}")
t14 = st[:, 14].astype(np.float64) / 100.0 - t0
print(f"wave 0 done -> all waves done     {q(t14 - t[:, 3])}")
print(f"table reduction + partial stores  {q(t[:, 4] - t14)}")
print(f"drain + arrival count             {q(t[:, 5] - t[:, 4])}")
print(f"unit combine (last arrivers only) {q((t[:, 6] - t[:, 5])[(t[:, 6] - t[:, 5]) > 0.5]) if ((t[:, 6] - t[:, 5]) > 0.5).any() else '-'}")
last = (t[:, 6] - t[:, 5]) > 0.5
if last.any():
    c = st[last][:, 8:14].astype(np.float64) / 100.0 - t0
    e5 = t[last, 5]
    print(f"  combine: tables initialised     {q(c[:, 0] - e5)}")
    print(f"  combine: loads + entries filed  {q(c[:, 1] - c[:, 0])}")
    print(f"  combine: partials staged        {q(c[:, 2] - c[:, 1])}")
    print(f"  combine: first batch folded     {q(c[:, 3] - c[:, 2])}")
    print(f"  combine: further batches        {q(c[:, 4] - c[:, 3])}")
    print(f"  combine: rows written           {q(c[:, 5] - c[:, 4])}")
    print(f"  combine: to exit                {q(t[last, 6] - c[:, 5])}")
print(f"exit (since first entry)          {q(t[:, 6])}")
print(f"tail: last loop end -> last exit  {end - t[:, 3].max():7.1f}")
xcc = (st[:, 7] >> 32) & 0xF
hw = st[:, 7] & 0xFFFFFFFF
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7 if False else (hw >> 13) & 0x3
print("per XCC: workgroups, median loop us, last exit us")
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print(f"  xcc {x}: {int(m.sum()):4d}  {np.median((t[:, 3] - t[:, 2])[m]):7.1f}  {t[m, 6].max():7.1f}")
# slowest / fastest workgroups
order = np.argsort(t[:, 3])
print("ten last workgroups to leave the loop: chunk, xcc, entry, loop us, loop end")
for i in order[-10:]:
    print(f"  {i:5d} xcc {int(xcc[i])} {t[i, 0]:7.1f} {t[i, 3] - t[i, 2]:7.1f} {t[i, 3]:7.1f}")
# occupancy over time: workgroups inside the loop per 10 us bucket
edges = np.arange(0, end + 10, 10)
inside = [(int(((t[:, 2] <= e) & (t[:, 3] > e)).sum())) for e in edges]
print("workgroups inside the loop every 10 us:", inside)
engine.__exit__(None, None, None)
