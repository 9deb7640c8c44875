"""Synthetic TPC-H-shaped lineitem, generated directly in HBM (bench + large parity tests).

Counter-based: value(row i, column c) = f(seed, c, i) (csrc/hs_ops.hip k_gen_lineitem; CPU twin in
oracle/q1_oracle.c), so any block can be produced independently on any rank.  Distributions follow
SURVEY.md section 8d.  The table is registered with the engine under a path that holds only a BlockFile
header (schema, zero blocks): planning reads the schema from it like from any table, the scan finds
the columns already resident.
"""

from __future__ import annotations

from pathlib import Path

import torch

from . import constants
from . import hipspark as hs
from .constants import ColumnType
from .device import DCol, Device
from .io import BlockFile
from .table import DeviceTable

SEED = 20251003
ROWS_SF1 = 6_001_215
LINEITEM_ROWS = {1: 6_001_215, 10: 59_986_052, 100: 600_037_902}

LINEITEM_SCHEMA = [
    ("l_orderkey", ColumnType.INTEGER),
    ("l_quantity", ColumnType.FLOAT),
    ("l_extendedprice", ColumnType.FLOAT),
    ("l_discount", ColumnType.FLOAT),
    ("l_tax", ColumnType.FLOAT),
    ("l_returnflag", ColumnType.STRING),
    ("l_shipdate", ColumnType.TIMESTAMP),
]
Q1_BYTES_PER_ROW = 26  # 4 x f32 + i64 + (1 length byte + 1 payload byte) of l_returnflag


def lineitem_rows(sf: float) -> int:
    return LINEITEM_ROWS.get(int(sf), int(round(sf * ROWS_SF1))) if sf == int(sf) else int(round(sf * ROWS_SF1))


def block_sizes(total_rows: int, rows_per_block: int | None = None) -> list[int]:
    per = rows_per_block or constants.ROWS_PER_BLOCK
    sizes = [per] * (total_rows // per)
    if total_rows % per:
        sizes.append(total_rows % per)
    return sizes


ORDERS_SCHEMA = [("o_orderkey", ColumnType.INTEGER), ("o_orderpriority", ColumnType.STRING)]


def orders_rows(lineitem_total_rows: int) -> int:
    """Four lineitems per order: sf=10 -> 14 996 513 orders."""
    return (lineitem_total_rows + 3) // 4


def strings_from_codes(dev: Device, codes: torch.Tensor, n: int, entries: list[str]) -> DCol:
    """A plain STRING column (length bytes + payload + offsets, as a BlockFile reader would deliver it) holding
    entries[codes[i]]: how the synthetic tables get their string columns - the engine then meets ordinary strings
    and does its own dictionary encoding at table open."""
    idx = dev.empty(n, torch.int64)
    idx.copy_(codes[:n])
    return dev.gather_col(dev.dict_column(tuple(e.encode() for e in entries)), idx, n)


def make_orders(dev: Device, path: Path, n_orders: int, seed: int = SEED, rank: int = 0, world: int = 1,
                rows_per_block: int | None = None) -> DeviceTable:
    """Synthetic orders of BASELINE config 4: o_orderkey = key(perm(row)) (every key once, build order != key order),
    o_orderpriority one of the five TPC-H strings (5-15 bytes); this rank's blocks (block b belongs to rank b % world).
    CPU twin: oracle/q45_oracle.c q4_gen_orders."""
    from .workloads import PRIORITIES  # noqa: PLC0415

    per = rows_per_block or constants.ROWS_PER_BLOCK
    sizes = block_sizes(n_orders, per)
    mine = [(b, n) for b, n in enumerate(sizes) if b % world == rank]
    n_local = sum(n for _, n in mine)
    okey = dev.empty(n_local, torch.int32)
    code = dev.empty(n_local, torch.uint8)
    off = 0
    for b, n in mine:
        hs.check(dev.lib.hs_gen_orders(dev.stream, seed, b * per, n, n_orders, okey[off:].data_ptr(), code[off:].data_ptr()),
                 "hs_gen_orders")
        off += n
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    BlockFile(path, list(ORDERS_SCHEMA)).write_rows([])  # header only
    table = DeviceTable(path, list(ORDERS_SCHEMA), [n for _, n in mine], {}, ())
    table.global_blocks = [b for b, _ in mine]
    table.total_blocks = len(sizes)
    table.columns[0] = DCol(hs.I32, okey, n_local)
    table.columns[1] = strings_from_codes(dev, code, n_local, PRIORITIES)
    return table


def make_lineitem(dev: Device, path: Path, total_rows: int, seed: int = SEED, rank: int = 0, world: int = 1,
                  rows_per_block: int | None = None, with_orderkey: bool = False, with_shipmode: bool = False) -> DeviceTable:
    """Generate this rank's blocks (block b belongs to rank b % world) of a ``total_rows`` lineitem."""
    per = rows_per_block or constants.ROWS_PER_BLOCK
    sizes = block_sizes(total_rows, per)
    mine = [(b, n) for b, n in enumerate(sizes) if b % world == rank]
    n_local = sum(n for _, n in mine)
    qty = dev.empty(n_local, torch.float32)
    price = dev.empty(n_local, torch.float32)
    disc = dev.empty(n_local, torch.float32)
    tax = dev.empty(n_local, torch.float32)
    ship = dev.empty(n_local, torch.int64)
    flag = dev.empty(n_local, torch.uint8)
    lens = dev.empty(n_local, torch.uint8)
    okey = dev.empty(n_local, torch.int32) if with_orderkey else None
    mode = dev.empty(n_local, torch.uint8) if with_shipmode else None
    off = 0
    for b, n in mine:
        hs.check(dev.lib.hs_gen_lineitem(
            dev.stream, seed, b * per, n, qty[off:].data_ptr(), price[off:].data_ptr(), disc[off:].data_ptr(),
            tax[off:].data_ptr(), ship[off:].data_ptr(), flag[off:].data_ptr(), lens[off:].data_ptr(),
            okey[off:].data_ptr() if okey is not None else None,
            mode[off:].data_ptr() if mode is not None else None), "hs_gen_lineitem")
        off += n
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    schema = list(LINEITEM_SCHEMA) + ([("l_shipmode", ColumnType.STRING)] if with_shipmode else [])
    BlockFile(path, schema).write_rows([])  # header only
    table = DeviceTable(path, schema, [n for _, n in mine], {}, ())
    table.global_blocks = [b for b, _ in mine]
    table.total_blocks = len(sizes)
    table.columns[1] = DCol(hs.F32, qty, n_local)
    table.columns[2] = DCol(hs.F32, price, n_local)
    table.columns[3] = DCol(hs.F32, disc, n_local)
    table.columns[4] = DCol(hs.F32, tax, n_local)
    table.columns[5] = DCol(hs.STR, flag, n_local, lens=lens, offs=None, fixed_len=1)
    table.columns[6] = DCol(hs.I64, ship, n_local)
    if okey is not None:
        table.columns[0] = DCol(hs.I32, okey, n_local)
    if mode is not None:
        from .workloads import SHIPMODES  # noqa: PLC0415

        table.columns[7] = strings_from_codes(dev, mode, n_local, SHIPMODES)
    return table
