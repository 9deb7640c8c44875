"""Table preparation next to the hot path (reference: src/mini_spark/utils.py:179-203).

``convert_csv_to_block_file`` turns a csv file with a header line into a BlockFile the engine scans - the
step the reference's benchmark runs before its queries (examples/benchmark.py:45-49).  Columnar here: a batch of
csv rows is converted column by column with numpy and appended with the reference's append-merge rule
(io.py:231-252), so the file is byte-identical to the one the reference's converter writes (pinned by
tests/golden/ingest.bin, written by the reference's own BlockFile writer).
"""

from __future__ import annotations

import csv
from datetime import datetime
from pathlib import Path

import numpy as np

from . import constants
from .constants import ColumnType, Schema
from .io import BlockFile, StrCol

_EPOCH = datetime(1970, 1, 1)


def _timestamp_us(text: str) -> int:
    delta = datetime.fromisoformat(text) - _EPOCH
    return (delta.days * 86_400 + delta.seconds) * 1_000_000 + delta.microseconds


def _convert(values: tuple[str, ...], col_type: ColumnType):
    if col_type == ColumnType.INTEGER:
        wide = np.array([int(v) for v in values], dtype=np.int64)
        if wide.size and (wide.max() > constants.MAX_INT or wide.min() < constants.MIN_INT):
            raise OverflowError("INTEGER value does not fit 32 bits")  # the reference's struct.pack('<i') fails too
        return wide.astype(np.int32)
    if col_type == ColumnType.FLOAT:
        wide = np.array([float(v) for v in values], dtype=np.float64)
        with np.errstate(over="ignore"):
            narrow = wide.astype(np.float32)
        if np.any(np.isinf(narrow) & ~np.isinf(wide)):
            raise OverflowError("float too large to pack with f format")
        return narrow
    if col_type == ColumnType.TIMESTAMP:
        return np.array([_timestamp_us(v) for v in values], dtype=np.int64)
    return StrCol.from_strings(list(values))


def convert_csv_to_block_file(csv_file: Path, block_file: Path, schema: Schema,
                              batch_size: int | None = None) -> None:
    """csv (first line = header, skipped; one field per schema column, in order) -> BlockFile.
    INTEGER = ``int(text)``, FLOAT = ``float(text)`` stored as f32, TIMESTAMP = ``datetime.fromisoformat``,
    STRING as is (ASCII, <= 255 bytes)."""
    csv_file, block_file = Path(csv_file), Path(block_file)
    if block_file.exists():
        raise FileExistsError(f"File {block_file} already exists")
    batch_size = batch_size or constants.ROWS_PER_BLOCK
    out = BlockFile(block_file, list(schema))
    wrote = False
    with csv_file.open(newline="") as f:
        reader = csv.reader(f)
        next(reader)  # header
        while True:
            rows = []
            for row in reader:
                if len(row) != len(schema):
                    raise ValueError(f"csv row has {len(row)} fields, the schema has {len(schema)}")
                rows.append(row)
                if len(rows) >= batch_size:
                    break
            if not rows:
                break
            columns = list(zip(*rows))
            out.append_raw([_convert(col, t) for col, (_, t) in zip(columns, schema)])
            wrote = True
    if not wrote:
        out.write_rows([])  # header-only csv: a BlockFile with the schema and no blocks
