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

# ---- the native reader (csrc/hs_engine.hip hs_table_open / hs_table_load): what a non-Python host gets ---------------------
import ctypes as C, json
from minispark_amd import hipspark as hs
lib = hs.load_library()
hip = C.CDLL("libamdhip64.so")
eng = C.c_void_p()
hs.check(lib.hs_engine_create(0, C.byref(eng)), "hs_engine_create")
ids = (C.c_int32 * len(q1_cols))(*q1_cols)
best = None
for rep in range(4):
    tab = C.c_void_p()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hs.check(lib.hs_table_open(eng, str(path).encode(), 0, 1, C.byref(tab)), "hs_table_open")
    hs.check(lib.hs_table_load(eng, tab, ids, len(q1_cols)), "hs_table_load")
    dt = time.perf_counter() - t0
    sec, nbytes = C.c_double(0), C.c_int64(0)
    lib.hs_engine_load_stats(eng, C.byref(sec), C.byref(nbytes))
    print(f"native ingest rep {rep}: open + load {dt*1e3:.0f} ms -> {want_bytes/dt/1e9:.2f} GB/s of referenced bytes; read + copy pipeline alone "
          f"{sec.value*1e3:.0f} ms = {nbytes.value/sec.value/1e9:.2f} GB/s ({nbytes.value} bytes moved)", flush=True)
    if best is None or dt < best[0]:
        best = (dt, sec.value, nbytes.value)
    if rep == 3:  # verify what landed in HBM against the generator
        col, n_rows = hs.hs_col(), C.c_int64(0)
        for cid, name, dtype in ((1, "l_extendedprice", np.float32), (5, "l_shipdate", np.int64)):
            hs.check(lib.hs_table_column(tab, cid, C.byref(col), C.byref(n_rows)), "hs_table_column")
            assert n_rows.value == rows
            host = np.empty(100000, dtype=dtype)
            assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(col.data), C.c_size_t(host.nbytes), 2) == 0
            assert np.array_equal(host, c[name][:100000]), name
        tail = np.empty(1000, dtype=np.float32)
        hs.check(lib.hs_table_column(tab, 0, C.byref(col), C.byref(n_rows)), "hs_table_column")
        assert hip.hipMemcpy(C.c_void_p(tail.ctypes.data), C.c_void_p(col.data + 4 * (rows - 1000)), C.c_size_t(4000), 2) == 0
        assert np.array_equal(tail, q1_native.gen(synth.SEED, rows - 1000, 1000)["l_quantity"])
        print("native reader: columns verified against the generator (head and tail)")
    lib.hs_table_close(tab)
lib.hs_engine_destroy(eng)
PCIE_PEAK = 63.0  # GB/s, PCIe Gen5 x16 per direction (spec)
dt, sec, nbytes = best
print(json.dumps({"metric": "BlockFile -> HBM ingest through the native reader (hs_table_open + hs_table_load), referenced bytes/sec",
                  "value": want_bytes / dt / 1e9, "unit": "GB/s", "sf": sf, "rows": rows, "file_bytes": path.stat().st_size,
                  "referenced_bytes": want_bytes, "pruned_fraction": 1 - want_bytes / path.stat().st_size,
                  "open_plus_load_ms": dt * 1e3, "pipeline_ms": sec * 1e3,
                  "roofline": {"bound": "pcie", "achieved": nbytes / sec / 1e9, "peak": PCIE_PEAK, "unit": "GB/s",
                               "frac": nbytes / sec / 1e9 / PCIE_PEAK,
                               "accounting": "bytes moved host -> device by the read + copy pipeline over its wall time (file in page cache)"}}))
import shutil; shutil.rmtree(scratch)
