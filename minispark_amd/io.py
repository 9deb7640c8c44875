"""BlockFile: the columnar on-disk format of the hot path (reference: src/mini_spark/io.py).

Byte format (all little-endian; reference writer io.py:47-60,74-109,217-229, reader io.py:63-71,112-170)::

    file   := header block* footer
    header := u8 ncols ; ncols x ( u8 type ; u8 name_len ; name bytes )
    block  := u32 nrows ; ncols x ( u64 payload_bytes ; payload )
      INTEGER   payload := nrows x i32          FLOAT payload := nrows x f32
      TIMESTAMP payload := nrows x i64 (us)     STRING payload := nrows x u8 length ; concatenated bytes
    footer := nblocks x u64 block_start ; u32 nblocks

This module is a from-scratch numpy codec for that format.  Two API levels:

* the *raw* level (``read_block_raw`` / ``write_raw_blocks`` / ``append_raw``) moves whole columns as
  numpy arrays (strings as ``StrCol(lens, data)``) - this is what the device upload path and the
  result writer use, and it supports column pruning (only the requested columns are read);
* the *row* level mirrors the reference's ``BlockFile`` methods name for name
  (``write_rows`` ... ``read_data_rows``) so callers of the reference keep working.

Quantisation happens here exactly as in the reference: FLOAT values are stored as IEEE binary32
(round-to-nearest-even, ``OverflowError`` if a finite double does not fit, like ``struct.pack('<f')``),
INTEGER values as i32 (``OverflowError`` when out of range, like ``int.to_bytes(4, signed=True)``).
"""

from __future__ import annotations

import itertools
import os
import struct
import time
from dataclasses import dataclass, field
from datetime import datetime
from pathlib import Path
from typing import Any, Iterable, Iterator, NamedTuple, Sequence

import numpy as np

from . import constants
from .constants import Columns, ColumnType, Row, Schema

MAX_COLUMNS = 0xFF
MAX_STR_LENGTH = 0xFF

_FIXED_DTYPES = {
    ColumnType.INTEGER: np.dtype("<i4"),
    ColumnType.FLOAT: np.dtype("<f4"),
    ColumnType.TIMESTAMP: np.dtype("<i8"),
}
_U8 = np.dtype(np.uint8)
_STRUCT_CODE = {ColumnType.INTEGER: ("i", 4), ColumnType.FLOAT: ("f", 4), ColumnType.TIMESTAMP: ("q", 8)}


class StrCol(NamedTuple):
    """A STRING column of one block: per-row byte lengths + the concatenated payload bytes."""

    lens: np.ndarray  # uint8[nrows]
    data: np.ndarray  # uint8[sum(lens)]

    def __len__(self) -> int:  # number of rows, so len(col) works for every raw column kind
        return int(self.lens.shape[0])

    def offsets(self) -> np.ndarray:
        off = np.zeros(len(self) + 1, dtype=np.int64)
        np.cumsum(self.lens, dtype=np.int64, out=off[1:])
        return off

    def to_list(self) -> list[str]:
        off = self.offsets()
        blob = self.data.tobytes()
        return [blob[off[i] : off[i + 1]].decode("utf-8") for i in range(len(self))]

    @staticmethod
    def from_strings(values: Sequence[str]) -> "StrCol":
        encoded = []
        for v in values:
            if type(v) is not str:
                raise AssertionError(f"STRING column holds {type(v).__name__}: {v!r}")
            b = v.encode("utf-8")
            if len(b) != len(v):
                # the reference writes len(chars) but reads that many bytes (io.py:101 vs :147) and so
                # corrupts the block; refuse instead of reproducing the corruption.
                raise ValueError(f"non-ASCII string not representable in a BlockFile: {v!r}")
            if len(b) > MAX_STR_LENGTH:
                raise ValueError(f"string longer than {MAX_STR_LENGTH} bytes: {v[:32]!r}...")
            encoded.append(b)
        lens = np.fromiter((len(b) for b in encoded), dtype=np.uint8, count=len(encoded))
        data = np.frombuffer(b"".join(encoded), dtype=np.uint8)
        return StrCol(lens, data)

    @staticmethod
    def concat(parts: Sequence["StrCol"]) -> "StrCol":
        return StrCol(np.concatenate([p.lens for p in parts]), np.concatenate([p.data for p in parts]))

    def slice(self, lo: int, hi: int) -> "StrCol":
        off = self.offsets()
        return StrCol(self.lens[lo:hi], self.data[off[lo] : off[min(hi, len(self))]])


RawColumn = Any  # np.ndarray (i4 / f4 / i8) or StrCol


def datetime_to_timestamp(dt: datetime) -> int:
    """Microseconds since the epoch of a naive *local* datetime (reference io.py:34-35)."""
    return int(dt.timestamp() * 1_000_000)


def timestamp_to_datetime(microseconds_since_epoch: int) -> datetime:
    """Inverse of :func:`datetime_to_timestamp` with the reference's float division (io.py:38-39)."""
    return datetime.fromtimestamp(microseconds_since_epoch / 1_000_000)


def timestamps_to_datetimes(us: np.ndarray) -> list[datetime]:
    """:func:`timestamp_to_datetime` for a whole column.  When the local time zone is UTC (``TZ=UTC``: the only setting
    under which the reference compares TIMESTAMPs consistently) the conversion is one vectorised step:
    ``fromtimestamp(v / 1e6)`` rounds the float seconds back to the same integer microseconds for every |v| < 4e15
    (year 2096), so the naive datetime is epoch + v.  Any other zone, or values out of that range: value by value."""
    if _zone_is_utc() and us.size and int(np.abs(us).max()) < 4_000_000_000_000_000:
        out = us.astype("datetime64[us]").tolist()
        if not out or isinstance(out[0], datetime):
            return out
    return [timestamp_to_datetime(v) for v in us.tolist()]


def _zone_is_utc() -> bool:
    """The process zone is UTC by NAME.  ``time.timezone == 0`` describes a zone's current standard offset only:
    Africa/Monrovia passes that test but was at -0:44:30 until 1972, and ``datetime.fromtimestamp`` applies the
    historical offset."""
    return time.tzname == ("UTC", "UTC") and time.timezone == 0 and time.daylight == 0


# --------------------------------------------------------------------------------------------------
# python values <-> raw numpy columns
# --------------------------------------------------------------------------------------------------


def quantise_float_column(values: np.ndarray) -> np.ndarray:
    """fp64 -> f32 exactly like ``struct.pack('<f', v)``: RNE, finite overflow is an error."""
    v64 = np.asarray(values, dtype=np.float64)
    with np.errstate(over="ignore", invalid="ignore"):
        v32 = v64.astype("<f4")
    if np.any(np.isfinite(v64) & ~np.isfinite(v32)):
        raise OverflowError("float too large to pack with f format")
    return v32


def quantise_int_column(values: Sequence[int] | np.ndarray) -> np.ndarray:
    """python ints -> i32, ``OverflowError`` when out of range (reference io.py:90)."""
    try:
        v64 = np.asarray(values, dtype=np.int64)
    except OverflowError:
        raise OverflowError("int too big to convert") from None
    if v64.size and (v64.min() < constants.MIN_INT or v64.max() > constants.MAX_INT):
        raise OverflowError("int too big to convert")
    return v64.astype("<i4")


def python_to_raw(values: Sequence[Any], col_type: ColumnType) -> RawColumn:
    if col_type == ColumnType.INTEGER:
        for v in values:
            if type(v) is not int:
                raise AssertionError(f"INTEGER column holds {type(v).__name__}: {v!r}")
        return quantise_int_column(list(values))
    if col_type == ColumnType.FLOAT:
        for v in values:
            if type(v) is not float:
                raise AssertionError(f"FLOAT column holds {type(v).__name__}: {v!r}")
        return quantise_float_column(np.array(values, dtype=np.float64))
    if col_type == ColumnType.TIMESTAMP:
        out = np.empty(len(values), dtype="<i8")
        for i, v in enumerate(values):
            dt = datetime.fromisoformat(v) if type(v) is str else v
            if type(dt) is not datetime:
                raise AssertionError(f"TIMESTAMP column holds {type(v).__name__}: {v!r}")
            out[i] = datetime_to_timestamp(dt)
        return out
    if col_type == ColumnType.STRING:
        return StrCol.from_strings(values)
    raise ValueError(f"Unsupported column type {col_type}")


def raw_to_python(col: RawColumn, col_type: ColumnType) -> list[Any]:
    if col_type == ColumnType.INTEGER:
        return np.asarray(col).astype(np.int64).tolist()
    if col_type == ColumnType.FLOAT:
        return np.asarray(col).astype(np.float64).tolist()
    if col_type == ColumnType.TIMESTAMP:
        return [timestamp_to_datetime(v) for v in np.asarray(col).tolist()]
    if col_type == ColumnType.STRING:
        return col.to_list()
    raise ValueError(f"Unsupported column type {col_type}")


def raw_len(col: RawColumn) -> int:
    return len(col)


def raw_concat(parts: Sequence[RawColumn]) -> RawColumn:
    if isinstance(parts[0], StrCol):
        return StrCol.concat(parts)
    return np.concatenate(parts)


def raw_slice(col: RawColumn, lo: int, hi: int) -> RawColumn:
    if isinstance(col, StrCol):
        return col.slice(lo, hi)
    return col[lo:hi]


# --------------------------------------------------------------------------------------------------
# header / footer / block codecs
# --------------------------------------------------------------------------------------------------


_SCHEMA_BYTES: dict[tuple, bytes] = {}


def encode_schema(schema: Schema) -> bytes:
    key = tuple(schema)
    hit = _SCHEMA_BYTES.get(key)
    if hit is None:
        hit = _SCHEMA_BYTES[key] = _encode_schema(schema)
    return hit


def _encode_schema(schema: Schema) -> bytes:
    if not len(schema) < MAX_COLUMNS:
        raise AssertionError("too many columns")
    out = bytearray([len(schema)])
    for name, col_type in schema:
        raw_name = name.encode("utf-8")
        if not len(name) < MAX_STR_LENGTH:
            raise AssertionError(f"column name too long: {name}")
        out += bytes([col_type.ordinal, len(name) & 0xFF]) + raw_name
    return bytes(out)


def decode_schema(buf: bytes) -> tuple[Schema, int]:
    """Parse the header at the start of ``buf``; returns (schema, header_size)."""
    ncols = buf[0]
    pos = 1
    schema: Schema = []
    for _ in range(ncols):
        col_type = ColumnType.from_ordinal(buf[pos])
        name_len = buf[pos + 1]
        name = bytes(buf[pos + 2 : pos + 2 + name_len]).decode("utf-8")
        schema.append((name, col_type))
        pos += 2 + name_len
    return schema, pos


def _deserialize_schema(f) -> Schema:  # name kept: the reference's tests call it (tests/test_io.py:27)
    head = f.read(1 + MAX_COLUMNS * (2 + MAX_STR_LENGTH))
    return decode_schema(head)[0]


def _column_bytes(col: np.ndarray, dtype: np.dtype) -> bytes:
    if col.dtype == dtype and col.flags.c_contiguous:
        return col.tobytes()
    return np.ascontiguousarray(col, dtype=dtype).tobytes()


def encode_block(schema: Schema, cols: Sequence[RawColumn]) -> bytes:
    nrows = raw_len(cols[0])
    parts = [struct.pack("<I", nrows)]
    for (_, col_type), col in zip(schema, cols, strict=True):
        if raw_len(col) != nrows:
            raise ValueError("ragged block: columns differ in length")
        if col_type == ColumnType.STRING:
            payload = _column_bytes(col.lens, _U8) + _column_bytes(col.data, _U8)
        else:
            payload = _column_bytes(col, _FIXED_DTYPES[col_type])
        parts.append(struct.pack("<Q", len(payload)))
        parts.append(payload)
    return b"".join(parts)


def encode_footer(block_starts: Sequence[int]) -> bytes:
    return struct.pack(f"<{len(block_starts)}QI", *block_starts, len(block_starts))


class BlockLayout(NamedTuple):
    """Where the columns of one block live in the file (absolute payload offsets)."""

    nrows: int
    spans: list[tuple[int, int]]  # per column: (payload_offset, payload_bytes)


@dataclass
class BlockFile:
    file: Path
    schema: Schema = field(default_factory=list)
    _block_starts: list[int] | None = field(default=None, init=False, repr=False)
    _file_schema: Schema | None = field(default=None, init=False, repr=False)

    def __post_init__(self) -> None:
        self.file = Path(self.file)

    # ---- metadata --------------------------------------------------------------------------------
    @property
    def block_starts(self) -> list[int]:
        if self._block_starts is None:
            with self.file.open("rb") as f:
                f.seek(-4, os.SEEK_END)
                nblocks = int(np.frombuffer(f.read(4), dtype="<u4")[0])
                f.seek(-4 - 8 * nblocks, os.SEEK_END)
                self._block_starts = np.frombuffer(f.read(8 * nblocks), dtype="<u8").astype(np.int64).tolist()
        return self._block_starts

    @property
    def file_schema(self) -> Schema:
        if self._file_schema is None:
            with self.file.open("rb") as f:
                self._file_schema = _deserialize_schema(f)
        return self._file_schema

    def _invalidate(self) -> None:
        self._block_starts = None
        self._file_schema = None

    def block_layout(self, block_id: int, f=None) -> BlockLayout:
        """Walk the u64 length prefixes of one block (ncols tiny seeks; no payload is read)."""
        own = f is None
        if own:
            f = self.file.open("rb")
        try:
            pos = self.block_starts[block_id]
            f.seek(pos)
            nrows = int(np.frombuffer(f.read(4), dtype="<u4")[0])
            pos += 4
            spans = []
            for _ in self.file_schema:
                f.seek(pos)
                nbytes = int(np.frombuffer(f.read(8), dtype="<u8")[0])
                spans.append((pos + 8, nbytes))
                pos += 8 + nbytes
            return BlockLayout(nrows, spans)
        finally:
            if own:
                f.close()

    def block_rows(self) -> list[int]:
        out = []
        with self.file.open("rb") as f:
            for start in self.block_starts:
                f.seek(start)
                out.append(int(np.frombuffer(f.read(4), dtype="<u4")[0]))
        return out

    def rows(self) -> int:
        return sum(self.block_rows())

    # ---- raw level -------------------------------------------------------------------------------
    def read_block_raw(self, block_id: int, col_ids: Sequence[int] | None = None, f=None) -> list[RawColumn]:
        """Read the requested columns of one block as numpy arrays (column pruning: only their bytes)."""
        schema = self.file_schema
        col_ids = list(range(len(schema))) if col_ids is None else list(col_ids)
        own = f is None
        if own:
            f = self.file.open("rb")
        try:
            layout = self.block_layout(block_id, f)
            out: list[RawColumn] = []
            for cid in col_ids:
                off, nbytes = layout.spans[cid]
                f.seek(off)
                buf = f.read(nbytes)
                col_type = schema[cid][1]
                if col_type == ColumnType.STRING:
                    arr = np.frombuffer(buf, dtype=np.uint8)
                    out.append(StrCol(arr[: layout.nrows], arr[layout.nrows :]))
                else:
                    out.append(np.frombuffer(buf, dtype=_FIXED_DTYPES[col_type]))
            return out
        finally:
            if own:
                f.close()

    def write_raw_blocks(self, schema: Schema, blocks: Iterable[Sequence[RawColumn]]) -> "BlockFile":
        """Write a whole file, one on-disk block per element of ``blocks`` (caller picks boundaries)."""
        if not schema:
            raise AssertionError("schema required")
        self.schema = list(schema)
        self._invalidate()
        starts = []
        header = encode_schema(schema)
        pos = len(header)
        pieces = [header]
        with self.file.open("wb") as f:
            for cols in blocks:
                if raw_len(cols[0]) == 0:
                    continue
                block = encode_block(schema, cols)
                starts.append(pos)
                pos += len(block)
                pieces.append(block)
                if pos > (64 << 20):  # flush large files piecewise, small ones with a single write
                    f.write(b"".join(pieces))
                    pieces = []
            pieces.append(encode_footer(starts))
            f.write(b"".join(pieces))
        return self

    def write_raw(self, schema: Schema, cols: Sequence[RawColumn]) -> "BlockFile":
        """Write columns, splitting into ROWS_PER_BLOCK blocks like the reference (io.py:81)."""
        return self.write_raw_blocks(schema, _split_rows(cols, constants.ROWS_PER_BLOCK))

    def append_raw(self, cols: Sequence[RawColumn]) -> "BlockFile":
        """Append with the reference's merge rule (io.py:231-252): top up the last block to
        ROWS_PER_BLOCK rows, then start new blocks; the footer is rewritten."""
        self._block_starts = None
        if not self.file.exists() or len(self.block_starts) == 0:
            return self.write_raw(self.schema, cols)
        schema = self.file_schema
        if self.schema != schema:
            raise AssertionError((self.file, self.schema, schema))
        starts = list(self.block_starts)
        with self.file.open("rb+") as f:
            last_rows = self.block_layout(len(starts) - 1, f).nrows
            if last_rows < constants.ROWS_PER_BLOCK:
                last = self.read_block_raw(len(starts) - 1, None, f)
                cols = [raw_concat([a, b]) for a, b in zip(last, cols, strict=True)]
                f.seek(starts.pop())
            else:
                f.seek(-(8 * len(starts) + 4), os.SEEK_END)
            for block in _split_rows(cols, constants.ROWS_PER_BLOCK):
                starts.append(f.tell())
                f.write(encode_block(schema, block))
            f.write(encode_footer(starts))
            f.truncate()
        self._block_starts = None
        return self

    # ---- row level (mirror of the reference's BlockFile methods) ---------------------------------------
    def write_data(self, data: Columns) -> "BlockFile":
        return self._write_python_columns(data, self.schema)

    def write_tuples(self, tuples: list[tuple[Any, ...]]) -> "BlockFile":
        return self._write_python_columns(tuple(map(list, zip(*tuples, strict=True))), self.schema)

    def write_rows(self, data: list[Row]) -> "BlockFile":
        if len(data) == 0:
            # empty table = header + u32 0 (a footer with zero blocks), reference io.py:206-211
            if self.schema:
                self._invalidate()
                self.file.write_bytes(encode_schema(self.schema) + encode_footer([]))
            return self
        self.schema = [(key, ColumnType.of(value)) for key, value in data[0].items()]
        columns = tuple([row[name] for row in data] for name, _ in self.schema)
        return self._write_python_columns(columns, self.schema)

    def _write_python_columns(self, columns: Columns, schema: Schema) -> "BlockFile":
        if not schema:
            raise AssertionError("schema required")
        raw = [python_to_raw(col, col_type) for col, (_, col_type) in zip(columns, schema, strict=True)]
        return self.write_raw(schema, raw)

    def append_data(self, data: Columns) -> "BlockFile":
        self._block_starts = None
        schema = self.schema if (not self.file.exists() or len(self.block_starts) == 0) else self.file_schema
        raw = [python_to_raw(col, col_type) for col, (_, col_type) in zip(data, schema, strict=True)]
        return self.append_raw(raw)

    def append_tuples(self, data: list[tuple[Any, ...]]) -> "BlockFile":
        return self.append_data(tuple(map(list, zip(*data, strict=True))))

    def append_rows(self, data: list[Row]) -> "BlockFile":
        if not self.schema:
            raise AssertionError("schema required")
        names = list(data[0].keys())
        return self.append_data(tuple([row[name] for row in data] for name in names))

    def read_block_data_columns_by_id(self, block_id: int, f=None) -> Columns:
        schema = self.file_schema
        raw = self.read_block_raw(block_id, None, f)
        return tuple(raw_to_python(col, col_type) for col, (_, col_type) in zip(raw, schema, strict=True))

    def read_block_data(self, block_id: int) -> list[tuple[Any, ...]]:
        return list(zip(*self.read_block_data_columns_by_id(block_id), strict=True))

    def read_block_data_columns_sequentially(self) -> Iterator[Columns]:
        with self.file.open("rb") as f:
            for block_id in range(len(self.block_starts)):
                yield self.read_block_data_columns_by_id(block_id, f)

    def read_blocks_sequentially(self) -> Iterator[list[Row]]:
        names = [name for name, _ in self.file_schema]
        for block_id in range(len(self.block_starts)):
            yield [dict(zip(names, row, strict=True)) for row in self.read_block_data(block_id)]

    def read_data_rows(self) -> Iterator[Row]:
        try:
            small = self.file.stat().st_size <= _SMALL_FILE
        except OSError:
            small = False
        if small:  # result files are a few hundred bytes: one read, parse in memory
            with open(self.file, "rb") as f:
                yield from _rows_from_bytes(f.read())
            return
        for block in self.read_blocks_sequentially():
            yield from block

    def merge_files(self, files: list[Path]) -> "BlockFile":
        self.schema = BlockFile(files[0]).file_schema
        for other in files:
            src = BlockFile(other)
            if src.file_schema != self.schema:
                raise AssertionError("schema mismatch in merge_files")
            for block_id in range(len(src.block_starts)):
                self.append_raw(src.read_block_raw(block_id))
        return self


_SMALL_FILE = 1 << 20


def write_single_block_file(path: Any, schema: Schema, cols: Sequence[RawColumn]) -> None:
    """header + one block + footer in a single write (result files: a handful of rows)."""
    header = encode_schema(schema)
    with open(path, "wb") as f:
        f.write(b"".join((header, encode_block(schema, cols), struct.pack("<QI", len(header), 1))))


def _rows_from_bytes(buf: bytes) -> Iterator[Row]:
    """Whole-file decode of a small BlockFile held in memory (same format walk as the file-based reader,
    struct-based: result files hold a handful of rows and numpy's per-call overhead would dominate)."""
    schema, _ = decode_schema(buf)
    names = [n for n, _ in schema]
    (nblocks,) = struct.unpack_from("<I", buf, len(buf) - 4)
    starts = struct.unpack_from(f"<{nblocks}Q", buf, len(buf) - 4 - 8 * nblocks)
    for start in starts:
        (nrows,) = struct.unpack_from("<I", buf, start)
        pos = start + 4
        cols = []
        for _, col_type in schema:
            (nbytes,) = struct.unpack_from("<Q", buf, pos)
            pos += 8
            if col_type == ColumnType.STRING:
                lens = buf[pos: pos + nrows]
                p = pos + nrows
                vals = []
                for ln in lens:
                    vals.append(buf[p: p + ln].decode("utf-8"))
                    p += ln
                cols.append(vals)
            else:
                code, size = _STRUCT_CODE[col_type]
                vals = struct.unpack_from(f"<{nbytes // size}{code}", buf, pos)
                cols.append([timestamp_to_datetime(v) for v in vals] if col_type == ColumnType.TIMESTAMP else vals)
            pos += nbytes
        for row in zip(*cols, strict=True):
            yield dict(zip(names, row, strict=True))


def python_columns(schema: Schema, cols: Sequence[RawColumn]) -> list[list]:
    """Raw columns -> lists of the Python values a reader of the BlockFile holding them would produce (f32 widened to
    float, TIMESTAMP as datetime, STRING as str): the column-wise form of rows_from_raw."""
    values = []
    for (_, col_type), c in zip(schema, cols, strict=True):
        if isinstance(c, StrCol):
            data = c.data.tobytes()
            vals, p = [], 0
            for ln in c.lens.tolist():
                vals.append(data[p: p + ln].decode("utf-8"))
                p += ln
            values.append(vals)
        elif col_type == ColumnType.TIMESTAMP:
            values.append(timestamps_to_datetimes(np.asarray(c, dtype=np.int64)))
        else:
            values.append(c.tolist())
    return values


class LazyRaw:
    """A SMALL result as it left the device: the Python values of its columns, decoded straight from the image bytes
    (`py_columns`: what rows are made of), and the raw numpy columns only when somebody asks for the file or the
    column-wise form (`build()`).  A query with a handful of result rows spends more host time wrapping them in numpy
    views than the finish launch takes to produce them."""

    __slots__ = ("py_columns", "_build", "_raw")

    def __init__(self, py_columns: list, build: Any) -> None:
        self.py_columns, self._build, self._raw = py_columns, build, None

    def build(self) -> list:
        if self._raw is None:
            self._raw = self._build()
        return self._raw


_ROW_BUILDERS: dict[int, Any] = {}


def _row_builder(width: int) -> Any:
    """names, columns -> [{name: value, ...} per row] as ONE list comprehension over a dict display of `width`
    entries: ~2x the speed of dict(zip(names, row)) per row - result sets of 10^5 groups spend their time here."""
    fn = _ROW_BUILDERS.get(width)
    if fn is None:
        keys = ", ".join(f"k{i}" for i in range(width))
        vals = ", ".join(f"v{i}" for i in range(width))
        cols = ", ".join(f"c{i}" for i in range(width))
        body = ", ".join(f"k{i}: v{i}" for i in range(width))
        comma = "," if width == 1 else ""
        src = (f"def build(names, columns):\n    {keys}{comma} = names\n    {cols}{comma} = columns\n"
               f"    return [{{{body}}} for {vals}{comma} in zip({cols})]\n")
        scope: dict[str, Any] = {}
        exec(src, scope)  # noqa: S102 - generated from an integer only
        fn = _ROW_BUILDERS[width] = scope["build"]
    return fn


def rows_list_from_raw(schema: Schema, cols: Sequence[RawColumn]) -> list[Row]:
    """rows_from_raw as a list, built without a generator in between."""
    names = [n for n, _ in schema]
    values = python_columns(schema, cols)
    if not names:
        return []
    if len({len(v) for v in values}) != 1:
        raise ValueError("columns of one result differ in length")
    # 10^5 new dicts would trigger a dozen passes of the cyclic collector over objects that cannot be garbage yet
    # (a third of the construction time): paused for the one comprehension
    import gc  # noqa: PLC0415

    paused = len(values[0]) >= 4096 and gc.isenabled()
    if paused:
        gc.disable()
    try:
        return _row_builder(len(names))(names, values)
    finally:
        if paused:
            gc.enable()


def rows_from_raw(schema: Schema, cols: Sequence[RawColumn]) -> Iterator[Row]:
    """Rows (dicts of Python values) straight from raw columns: the values a reader of the BlockFile
    holding these columns would produce (f32 widened to float, TIMESTAMP as datetime, STRING as str)."""
    yield from rows_list_from_raw(schema, cols)


def _split_rows(cols: Sequence[RawColumn], rows_per_block: int) -> Iterator[list[RawColumn]]:
    total = raw_len(cols[0]) if len(cols) else 0
    str_offsets = {i: c.offsets() for i, c in enumerate(cols) if isinstance(c, StrCol)}
    for lo in range(0, total, rows_per_block):
        hi = min(lo + rows_per_block, total)
        block = []
        for i, c in enumerate(cols):
            if isinstance(c, StrCol):
                off = str_offsets[i]
                block.append(StrCol(c.lens[lo:hi], c.data[off[lo] : off[hi]]))
            else:
                block.append(c[lo:hi])
        yield block
