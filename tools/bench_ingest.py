"""BlockFile -> HBM ingest rate: write a lineitem BlockFile with the Q1 columns + a wide
filler column (so pruning matters), then time open_table + load_columns of the Q1 columns.
Usage: python tools/bench_ingest.py [sf]   (file goes to /dev/shm: page-cache speed, i.e. the PCIe-side bound)"""
import os, sys, tempfile, time
from pathlib import Path
os.environ.setdefault("TZ", "UTC"); time.tzset()
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from minispark_amd import synth, table as tbl
from minispark_amd.constants import ColumnType as T
from minispark_amd.device import Device
from minispark_amd.io import BlockFile, StrCol
from oracle import q1_native

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
rows = synth.lineitem_rows(sf)
dev = Device(0)
scratch = Path(tempfile.mkdtemp(prefix="hs_ingest_", dir="/dev/shm"))
path = scratch / "lineitem.bin"
schema = [("l_quantity", T.FLOAT), ("l_extendedprice", T.FLOAT), ("l_discount", T.FLOAT), ("l_tax", T.FLOAT),
          ("l_returnflag", T.STRING), ("l_shipdate", T.TIMESTAMP), ("l_comment", T.STRING)]
per = 2 * 1024 * 1024
def blocks():
    for lo in range(0, rows, per):
        n = min(per, rows - lo)
        c = q1_native.gen(synth.SEED, lo, n)
        filler = StrCol(np.full(n, 27, np.uint8), np.full(n * 27, ord("x"), np.uint8))
        yield [c["l_quantity"], c["l_extendedprice"], c["l_discount"], c["l_tax"],
               StrCol(np.ones(n, np.uint8), c["l_returnflag"]), c["l_shipdate"], filler]
t0 = time.perf_counter()
BlockFile(path).write_raw_blocks(schema, blocks())
print(f"wrote {path.stat().st_size/1e9:.2f} GB ({rows} rows) in {time.perf_counter()-t0:.1f} s")
q1_cols = [0, 1, 2, 3, 4, 5]
want_bytes = rows * 26
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = tbl.open_table(path)
    tbl.load_columns(dev, t, q1_cols)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"ingest rep {rep}: {dt*1e3:.0f} ms  -> {want_bytes/dt/1e9:.2f} GB/s of referenced bytes "
          f"({path.stat().st_size/dt/1e9:.2f} GB/s of file bytes avoided by pruning: {1 - want_bytes/path.stat().st_size:.0%} skipped)")
c = q1_native.gen(synth.SEED, 0, min(rows, 100000))
assert np.array_equal(t.columns[1].data[:100000].cpu().numpy(), c["l_extendedprice"][:100000])
assert t.columns[4].fixed_len == 1 and np.array_equal(t.columns[5].data[:100000].cpu().numpy(), c["l_shipdate"][:100000])
print("ingested columns verified against the generator")
import shutil; shutil.rmtree(scratch)
