"""BlockFile -> HBM: the device-side L0 reader (reference: io.py:112-163 ``_deserialize_block``,
zig block_file.zig:225-306).

Differences from the reference, none of which change results:

* **column pruning** - only the byte spans of the requested columns are read from disk and uploaded
  (the reference decodes all columns of a block, io.py:156-163);
* columns are stored contiguously across blocks, one buffer per column; the file's block boundaries
  are kept as ``unit_rows`` because a block is the unit of partial aggregation;
* STRING columns get their byte offsets from a device prefix sum, and ``fixed_len`` when every row has
  the same length (then no offsets are materialised).
"""

from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path

import numpy as np
import torch

from . import hipspark as hs
from .constants import ColumnType, Schema
from .device import FILE_KIND, DBatch, DCol, Device
from .io import BlockFile

_NP_DTYPE = {ColumnType.INTEGER: np.dtype("<i4"), ColumnType.FLOAT: np.dtype("<f4"),
             ColumnType.TIMESTAMP: np.dtype("<i8")}
_TORCH = {ColumnType.INTEGER: torch.int32, ColumnType.FLOAT: torch.float32, ColumnType.TIMESTAMP: torch.int64}


@dataclass
class DeviceTable:
    """Columns of one BlockFile resident in HBM (loaded lazily, per column)."""

    path: Path
    schema: Schema
    block_rows: list[int]
    columns: dict[int, DCol] = field(default_factory=dict)
    stamp: tuple = ()
    global_blocks: list[int] | None = None  # multi-GPU: file block ids of the local blocks
    total_blocks: int | None = None

    @property
    def nrows(self) -> int:
        return sum(self.block_rows)

    @property
    def unit_rows(self) -> list[int]:
        out = [0]
        for r in self.block_rows:
            out.append(out[-1] + r)
        return out


def file_stamp(path: Path) -> tuple:
    st = path.stat()
    return (st.st_mtime_ns, st.st_size)


def open_table(path: Path, rank: int = 0, world: int = 1) -> DeviceTable:
    """Open a BlockFile; with world > 1 this rank owns the blocks b with b % world == rank."""
    bf = BlockFile(path)
    rows = bf.block_rows()
    if world == 1:
        return DeviceTable(Path(path), list(bf.file_schema), rows, {}, file_stamp(Path(path)))
    mine = [b for b in range(len(rows)) if b % world == rank]
    return DeviceTable(Path(path), list(bf.file_schema), [rows[b] for b in mine], {}, file_stamp(Path(path)),
                       global_blocks=mine, total_blocks=len(rows))


def load_columns(dev: Device, table: DeviceTable, col_ids: list[int]) -> None:
    """Read the byte spans of ``col_ids`` block by block and place them in per-column device buffers."""
    missing = [c for c in col_ids if c not in table.columns]
    if not missing:
        return
    bf = BlockFile(table.path)
    nblocks = len(table.block_rows)
    total_rows = table.nrows
    file_blocks = table.global_blocks if table.global_blocks is not None else list(range(nblocks))
    with table.path.open("rb") as f:
        layouts = [bf.block_layout(b, f) for b in file_blocks]
        for cid in missing:
            col_type = table.schema[cid][1]
            if col_type == ColumnType.STRING:
                payload = sum(layouts[b].spans[cid][1] - layouts[b].nrows for b in range(nblocks))
                lens = dev.empty(total_rows, torch.uint8)
                data = dev.empty(payload, torch.uint8)
                row, byte = 0, 0
                for b in range(nblocks):
                    off, nbytes = layouts[b].spans[cid]
                    n = layouts[b].nrows
                    f.seek(off)
                    buf = np.frombuffer(f.read(nbytes), dtype=np.uint8)
                    lens[row: row + n].copy_(torch.from_numpy(buf[:n].copy()))
                    if nbytes > n:
                        data[byte: byte + nbytes - n].copy_(torch.from_numpy(buf[n:].copy()))
                    row += n
                    byte += nbytes - n
                table.columns[cid] = dev.string_col(lens, data, total_rows)
            else:
                out = dev.empty(total_rows, _TORCH[col_type])
                row = 0
                for b in range(nblocks):
                    off, nbytes = layouts[b].spans[cid]
                    n = layouts[b].nrows
                    f.seek(off)
                    buf = np.frombuffer(f.read(nbytes), dtype=_NP_DTYPE[col_type])
                    if len(buf) != n:
                        raise ValueError(f"{table.path}: block {b} column {cid} holds {len(buf)} values, expected {n}")
                    out[row: row + n].copy_(torch.from_numpy(buf.copy()))
                    row += n
                table.columns[cid] = DCol(FILE_KIND[col_type], out, total_rows)


def table_batch(table: DeviceTable, col_ids: list[int], alias: str = "") -> DBatch:
    prefix = f"{alias}." if alias else ""
    schema = [(prefix + table.schema[c][0], table.schema[c][1]) for c in col_ids]
    return DBatch(schema, [table.columns[c] for c in col_ids], table.nrows, table.unit_rows,
                  unit_ids=table.global_blocks, total_units=table.total_blocks)
