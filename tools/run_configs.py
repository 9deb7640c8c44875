"""BASELINE.json configs 4 and 5 at scale on one GPU, checked against independent numpy statements:
  config 4: orders JOIN lineitem ON l_orderkey = o_orderkey GROUP BY o_orderpriority
  config 5: STRING-key GROUP BY (CONCAT) with a LIKE predicate
Usage: python tools/run_configs.py [lineitem_rows]"""
import os, sys, tempfile, time
from pathlib import Path
os.environ.setdefault("TZ", "UTC"); time.tzset()
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from minispark_amd import constants, hipspark as hs, synth
from minispark_amd.constants import ColumnType as T
from minispark_amd.dataframe import DataFrame
from minispark_amd.device import DCol
from minispark_amd.execution import HipExecutionEngine
from minispark_amd.io import BlockFile
from minispark_amd.sql import Col, Functions as F
from minispark_amd.table import DeviceTable
from oracle import q1_native
from minispark_amd.workloads import PRIORITIES, SHIPMODES

n_li = int(sys.argv[1]) if len(sys.argv) > 1 else 6_001_215
n_ord = n_li // 4 + 1
scratch = Path(tempfile.mkdtemp(prefix="hs_cfg_", dir="/dev/shm"))
constants.SHUFFLE_FOLDER = scratch / "shuffle"
engine = HipExecutionEngine(0)
dev = engine.dev

# ---- lineitem with l_orderkey and a variable-length l_shipmode string column (host-built from the codes)
li_schema = [("l_orderkey", T.INTEGER), ("l_quantity", T.FLOAT), ("l_extendedprice", T.FLOAT), ("l_returnflag", T.STRING),
             ("l_shipmode", T.STRING)]
cols = q1_native.gen(synth.SEED, 0, n_li, orderkey=True, shipmode=True)
modes = np.array([m.encode() for m in SHIPMODES], dtype=object)
mode_len = np.array([len(m) for m in SHIPMODES], dtype=np.uint8)
codes = cols["l_shipmode_code"]
ship_lens = mode_len[codes]
ship_data = np.frombuffer(b"".join(modes[codes].tolist()), dtype=np.uint8)
li_path = scratch / "lineitem.bin"
BlockFile(li_path, li_schema).write_rows([])
li = DeviceTable(li_path, li_schema, synth.block_sizes(n_li), {}, ())
li.columns[0] = dev.fixed_col(hs.I32, cols["l_orderkey"])
li.columns[1] = dev.fixed_col(hs.F32, cols["l_quantity"])
li.columns[2] = dev.fixed_col(hs.F32, cols["l_extendedprice"])
li.columns[3] = DCol(hs.STR, dev.to_device(cols["l_returnflag"], torch.uint8), n_li, lens=dev.const_lens(1, n_li), fixed_len=1)
li.columns[4] = dev.string_col(dev.to_device(ship_lens, torch.uint8), dev.to_device(ship_data, torch.uint8), n_li)
engine.attach_device_table(li_path, li)

# ---- orders: key(j) for a seeded permutation, priority string
rng = np.random.default_rng(7)
perm = rng.permutation(n_ord)
okey = (32 * (perm // 8) + (perm % 8) + 1).astype(np.int32)
prio = rng.integers(0, 5, n_ord)
prio_b = np.array([p.encode() for p in PRIORITIES], dtype=object)
ord_schema = [("o_orderkey", T.INTEGER), ("o_orderpriority", T.STRING)]
ord_path = scratch / "orders.bin"
BlockFile(ord_path, ord_schema).write_rows([])
orders = DeviceTable(ord_path, ord_schema, synth.block_sizes(n_ord), {}, ())
orders.columns[0] = dev.fixed_col(hs.I32, okey)
orders.columns[1] = dev.string_col(dev.to_device(np.array([len(p) for p in PRIORITIES], np.uint8)[prio], torch.uint8),
                                   dev.to_device(np.frombuffer(b"".join(prio_b[prio].tolist()), np.uint8), torch.uint8), n_ord)
engine.attach_device_table(ord_path, orders)

def timed(build, reps=3):
    rows = None
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rows = build().collect()
        ts.append(time.perf_counter() - t0)
    return rows, min(ts)

# ---- config 4
def q4():
    o = DataFrame(engine).table(str(ord_path)).select(Col("o_orderkey"), Col("o_orderpriority"))
    l = DataFrame(engine).table(str(li_path)).select(Col("l_orderkey"), Col("l_quantity"))
    return (o.join(l, on=Col("o_orderkey") == Col("l_orderkey"), how="inner").group_by(Col("o_orderpriority"))
            .agg(F.count().alias("n"), F.sum(Col("l_quantity")).alias("qty")))
rows4, t4 = timed(q4)
prio_of_key = {}
order_prio = np.empty(int(okey.max()) + 1, dtype=np.int8); order_prio[:] = -1; order_prio[okey] = prio
lp = order_prio[cols["l_orderkey"]]
assert (lp >= 0).all()
want_n = np.bincount(lp, minlength=5)
want_q = np.bincount(lp, weights=cols["l_quantity"].astype(np.float64), minlength=5)
got = {r["o_orderpriority"]: r for r in rows4}
for i, p in enumerate(PRIORITIES):
    assert got[p]["n"] == int(want_n[i]), (p, got[p]["n"], want_n[i])
    assert abs(got[p]["qty"] - want_q[i]) <= 4e-7 * want_q[i], (p, got[p]["qty"], want_q[i])  # 10 partition partials, each f32
print(f"config4 join+groupby: orders {n_ord} x lineitem {n_li}: {t4*1e3:.1f} ms  ({n_li/t4/1e6:.0f} M probe rows/s)  counts exact, sums within f32 partial rounding")

# ---- config 5
def q5():
    return (DataFrame(engine).table(str(li_path)).filter(Col("l_shipmode").like("%AIR%"))
            .select((Col("l_returnflag") + "-" + Col("l_shipmode")).alias("k"), Col("l_quantity"))
            .group_by(Col("k")).agg(F.sum(Col("l_quantity")).alias("qty"), F.count()))
rows5, t5 = timed(q5)
air = np.isin(codes, [SHIPMODES.index("REG AIR"), SHIPMODES.index("AIR")])
flags = cols["l_returnflag"]
want = {}
for fl in (ord("A"), ord("N"), ord("R")):
    for m in ("REG AIR", "AIR"):
        sel = air & (flags == fl) & (codes == SHIPMODES.index(m))
        want[f"{chr(fl)}-{m}"] = (int(sel.sum()), float(cols["l_quantity"][sel].astype(np.float64).sum()))
got5 = {r["k"]: r for r in rows5}
assert set(got5) == set(want), (sorted(got5), sorted(want))
for k, (cnt, qty) in want.items():
    assert got5[k]["count"] == cnt and abs(got5[k]["qty"] - qty) <= 3e-7 * qty, (k, got5[k], cnt, qty)
print(f"config5 LIKE + CONCAT key group-by: lineitem {n_li}: {t5*1e3:.1f} ms  ({n_li/t5/1e6:.0f} M rows/s)  counts exact")
engine.__exit__(None, None, None)
