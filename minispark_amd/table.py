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
from typing import Any

import numpy as np
import torch

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

    def stored_column(self, cid: int) -> DCol:
        """Column ``cid`` as the file stores it (a dictionary-coded string column gives its plain form)."""
        col = self.columns[cid]
        return col.plain if col.plain is not None else col

    @property
    def unit_rows(self) -> list[int]:
        out = [0]
        for r in self.block_rows:
            out.append(out[-1] + r)
        return out


def file_stamp(path: Path) -> tuple:
    st = path.stat()
    return (st.st_mtime_ns, st.st_size)


def open_table(path: Path, rank: int = 0, world: int = 1, distributed: bool = False) -> DeviceTable:
    """Open a BlockFile; a distributed engine's rank owns the blocks b with b % world == rank (also at world 1:
    the multi-rank path then runs with one rank, which is how the RCCL collectives are tested on a one-GPU box)."""
    bf = BlockFile(path)
    rows = bf.block_rows()
    if world == 1 and not distributed:
        return DeviceTable(Path(path), list(bf.file_schema), rows, {}, file_stamp(Path(path)))
    mine = [b for b in range(len(rows)) if b % world == rank]
    return DeviceTable(Path(path), list(bf.file_schema), [rows[b] for b in mine], {}, file_stamp(Path(path)),
                       global_blocks=mine, total_blocks=len(rows))


def referenced_block_bytes(table: DeviceTable, col_ids: list[int]) -> list[int]:
    """Bytes the listed columns occupy in each of the table's (local) blocks - what a scan has to bring into HBM."""
    bf = BlockFile(table.path)
    blocks = table.global_blocks if table.global_blocks is not None else list(range(len(table.block_rows)))
    with table.path.open("rb") as f:
        return [sum(bf.block_layout(b, f).spans[c][1] for c in col_ids) for b in blocks]


def sub_table(table: DeviceTable, local_blocks: list[int]) -> DeviceTable:
    """The table restricted to some of its local blocks (columns not loaded): one range of a streamed scan."""
    file_blocks = table.global_blocks if table.global_blocks is not None else list(range(len(table.block_rows)))
    return DeviceTable(table.path, list(table.schema), [table.block_rows[b] for b in local_blocks], {}, table.stamp,
                       global_blocks=[file_blocks[b] for b in local_blocks],
                       total_blocks=table.total_blocks if table.total_blocks is not None else len(table.block_rows))


# ---- BlockFile -> HBM ingest ---------------------------------------------------------
# Column pruning first (only the referenced columns' byte spans are read), then the library's reader pipeline
# (hs_read_spans): reader threads pread chunks straight into pinned staging slots while the H2D copies of earlier chunks
# are in flight.  Disk / page cache and PCIe run concurrently; nothing is copied twice on the host.
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

    # destination buffers + the list of (file offset, nbytes, destination byte view) pieces
    pieces: list[tuple[int, int, torch.Tensor]] = []
    finish: list[tuple[int, torch.Tensor, torch.Tensor | None]] = []
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
                if n:
                    pieces.append((off, n, lens[row: row + n]))
                if nbytes > n:
                    pieces.append((off + n, nbytes - n, data[byte: byte + nbytes - n]))
                row += n
                byte += nbytes - n
            finish.append((cid, data, lens))
        else:
            out = dev.empty(total_rows, _TORCH[col_type])
            esize = out.element_size()
            row = 0
            for b in range(nblocks):
                off, nbytes = layouts[b].spans[cid]
                n = layouts[b].nrows
                if nbytes != n * esize:
                    raise ValueError(f"{table.path}: block {b} column {cid} holds {nbytes} bytes, expected {n * esize}")
                if n:
                    pieces.append((off, nbytes, out[row: row + n].view(torch.uint8)))
                row += n
            finish.append((cid, out, None))

    if pieces:
        # the read + host-to-device pipeline is the library's (csrc/hs_engine.hip hs_read_spans: reader threads pread
        # chunks into pinned staging while earlier chunks' copies are in flight) - round 2 ran its own copy of that
        # pipeline here, in Python threads, at 27 GB/s
        import ctypes as C  # noqa: PLC0415

        from . import hipspark as hs  # noqa: PLC0415

        spans = (hs.hs_span * len(pieces))()
        for i, (off, nbytes, dst) in enumerate(pieces):
            spans[i].file_offset, spans[i].bytes, spans[i].dst = off, nbytes, dst.data_ptr()
        torch.cuda.current_stream(dev.device).synchronize()  # the destinations were allocated on this stream
        if dev.rec is not None:
            dev.rec.poisoned = True  # file contents flow in: not a run to replay
        hs.check(dev._raw_lib.hs_read_spans(dev.native_engine(), str(table.path).encode(), spans, len(pieces)), "hs_read_spans")

    for cid, data, lens in finish:
        if lens is not None:
            table.columns[cid] = dev.string_col(lens, data, total_rows)
        else:
            table.columns[cid] = DCol(FILE_KIND[table.schema[cid][1]], data, total_rows)


def table_batch(table: DeviceTable, col_ids: list[int], alias: str = "") -> DBatch:
    prefix = f"{alias}." if alias else ""
    schema = [(prefix + table.schema[c][0], table.schema[c][1]) for c in col_ids]
    return DBatch(schema, [table.columns[c] for c in col_ids], table.nrows, table.unit_rows,
                  unit_ids=table.global_blocks, total_units=table.total_blocks)
