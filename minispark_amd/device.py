"""Device-resident columns / batches and the operator calls on them.

torch supplies device memory, the current HIP stream and (in :mod:`minispark_amd.distributed`) the
RCCL process group - plumbing only.  Every operator below is one or a few calls into libhipspark.so;
there is no torch arithmetic on the data path and no CPU fallback.

Layout in HBM: one contiguous buffer per referenced column spanning all rows of the
batch; STRING columns are ``lens`` (u8 per row) + ``data`` (payload bytes) + ``offs`` (i64 exclusive
offsets, omitted when every row has the same length - ``fixed_len``).  Buffers are over-allocated by
``PAD`` bytes so the 16-byte row-quad loads of the fused kernels may run past the last row.
"""

from __future__ import annotations

import ctypes as C
import os
import struct
import time
from dataclasses import dataclass, field
from typing import Any, Sequence

import numpy as np
import torch

from . import hipspark as hs
from .constants import ColumnType, Schema
from .io import LazyRaw, StrCol, timestamp_to_datetime
from .lowering import NeedsDecoded, ProgramBuilder, StringParts, lower_aggregate

PAD = 64  # bytes of slack behind every buffer

def _strcol_bytes(col: StrCol) -> list[bytes]:
    data, out, pos = col.data.tobytes(), [], 0
    for ln in col.lens.tolist():
        out.append(data[pos: pos + ln])
        pos += ln
    return out


SMALL_RESULT_ROWS = 16  # results up to this many rows are decoded with struct straight from the result image
_IMAGE_HEADER = struct.Struct("<IIq")  # flags, "done" word, row count
_SMALL_STRUCTS: dict[tuple[int, int], Any] = {}


def _small_struct(kind: int, n: int) -> Any:
    st = _SMALL_STRUCTS.get((kind, n))
    if st is None:
        code = {hs.I32: "i", hs.F32: "f", hs.I64: "q", hs.F64: "d"}[kind]
        st = _SMALL_STRUCTS[(kind, n)] = struct.Struct(f"<{n}{code}")
    return st


def _decode_codes(codes: np.ndarray, entries: tuple) -> StrCol:
    """Host side of the result hand-over: code bytes -> the raw STRING column a BlockFile holds."""
    lens_of = np.array([len(e) for e in entries], dtype=np.uint8)
    codes = np.asarray(codes, dtype=np.uint8)
    return StrCol(lens_of[codes], np.frombuffer(b"".join(entries[c] for c in codes.tolist()), dtype=np.uint8).copy())


_TORCH_DTYPE = {hs.I32: torch.int32, hs.F32: torch.float32, hs.I64: torch.int64, hs.F64: torch.float64,
                hs.U8: torch.uint8}
_NP_DTYPE = {hs.I32: np.int32, hs.F32: np.float32, hs.I64: np.int64, hs.F64: np.float64, hs.U8: np.uint8}
FILE_KIND = {ColumnType.INTEGER: hs.I32, ColumnType.FLOAT: hs.F32, ColumnType.TIMESTAMP: hs.I64,
             ColumnType.STRING: hs.STR}


@dataclass
class DCol:
    kind: int
    data: torch.Tensor
    n: int
    lens: torch.Tensor | None = None
    offs: torch.Tensor | None = None
    fixed_len: int = -1
    # Dictionary-coded STRING column: `data` holds ONE CODE BYTE per row - to every kernel a string
    # column of fixed length 1, so GROUP BY / gathers / the exchange slabs work on it unchanged - and `dict` the
    # strings the codes stand for.  Whatever looks INSIDE the strings (LIKE, comparisons, concatenation, a join or
    # partition key, the result hand-over) goes through the lowering's dictionary forms or decodes first.
    dict: tuple | None = None
    plain: "DCol | None" = None  # the same rows as real strings, where they exist anyway (table columns)
    # hs.JOIN8_CODE / hs.JOIN8_UNIT: a VIRTUAL column of the fused join + aggregate - `data` is the probe
    # side's key column and the row's value is looked up in the join's byte table while the aggregate scans; only
    # Device.aggregate_join8 understands it
    virtual: int = 0

    def as_hs(self) -> hs.hs_col:
        c = hs.hs_col()
        c.kind = self.virtual or self.kind
        c.fixed_len = self.fixed_len
        c.data = self.data.data_ptr()
        c.lens = self.lens.data_ptr() if self.lens is not None else None
        c.offs = self.offs.data_ptr() if self.offs is not None else None
        return c

    def nbytes(self) -> int:
        total = self.data.numel() * self.data.element_size()
        if self.lens is not None:
            total += self.lens.numel()
        return total


@dataclass
class DBatch:
    """Columns + row count + unit boundaries.

    ``nrows`` is exact unless ``nrows_dev`` is set: then it is an upper bound (buffers are that large) and
    the exact count lives on the device in ``nrows_dev[0]`` (int64).  Operators pass that pointer to the
    kernels, so a query can run end to end without the host learning intermediate sizes."""

    schema: Schema
    cols: list[DCol]
    nrows: int
    unit_rows: list[int] | None = field(default_factory=list)  # host copy of the unit boundaries [n_units+1]
    nrows_dev: torch.Tensor | None = None
    unit_ids: list[int] | None = None     # multi-GPU: global id (file block id) of every local unit
    total_units: int | None = None        # multi-GPU: number of units over all ranks
    order: torch.Tensor | None = None     # per-row order key for the final merge (global unit id; <0 = padding)
    slab: torch.Tensor | None = None      # multi-GPU: the exchange slab the columns live in
    slab_layout: Any = None
    slab_cols: list[int] | None = None    # slab column holding each batch column (aggregates may share one)
    partitioned: bool = False             # multi-GPU: every rank holds DIFFERENT rows (else replicated)
    tail: dict | None = None              # partial rows emitted in slab form for aggregate_finish (the short tail)
    unit_col: torch.Tensor | None = None  # COMPUTED units (probe side of a join, rows left in place): u8 unit id per
    n_unit_ids: int = 0                   # row (0xff = dropped) and the number of units; unit_rows is then [0, nrows]
    join_task_id: Any = None              # the join that produced the computed units (for the fall-back decision)
    join8: dict | None = None             # the join's byte table + probe key column (Device.join8_table): the probe runs
                                          # INSIDE the partial aggregate (Device.aggregate_join8); such a batch carries
                                          # virtual columns and can feed nothing else

    def __post_init__(self) -> None:
        if self.nrows_dev is None and not self.unit_rows:
            self.unit_rows = [0, self.nrows]  # exact batch without explicit units: one unit

    @property
    def kinds(self) -> list[int]:
        return [c.kind for c in self.cols]

    @property
    def dicts(self) -> list:
        return [c.dict for c in self.cols]

    @property
    def lazy(self) -> bool:
        return self.nrows_dev is not None

    @property
    def n_units(self) -> int:
        return len(self.unit_rows) - 1 if self.unit_rows is not None else 1

    @property
    def n_dev_ptr(self):
        return self.nrows_dev.data_ptr() if self.nrows_dev is not None else None

    def column_index(self, name: str) -> int:
        for i, (col_name, _) in enumerate(self.schema):
            if col_name == name:
                return i
        raise ValueError(f'Column "{name}" not found in schema {self.schema}')


class TierExceeded(NotImplementedError):
    """The on-chip (LDS) aggregation tier cannot hold this launch: the engine switches to the global tier."""


class SlabUnsupported(NotImplementedError):
    """This partial aggregate cannot be written into a fixed-size exchange slab (variable-length string key):
    the engine exchanges its rows with the generic all-to-all instead."""


class RetryWithLargerDictionary(Exception):
    """A launch met more distinct group keys than its dictionary capacity (noticed at the query's single
    host round trip); the engine re-runs the query with larger capacities.  ``flags`` says which: HS_FLAG_DICT_FULL =
    the per-unit dictionaries of the partial aggregate, HS_FLAG_MERGE_FULL = the final merge's."""

    def __init__(self, flags: int = hs.FLAG_DICT_FULL | hs.FLAG_MERGE_FULL) -> None:
        super().__init__(f"dictionary capacity exceeded (flags {flags:#x})")
        self.unit_full = bool(flags & hs.FLAG_DICT_FULL)
        self.merge_full = bool(flags & hs.FLAG_MERGE_FULL)


class DeviceError(RuntimeError):
    pass


class Recording:
    """The device work of one query run, with every buffer it touches kept alive.  Library calls made while recording
    are captured INSIDE the library as the launches they perform (include/hipspark.h hs_capture_*): a run of
    consecutive library calls becomes one ``hs_capture_replay(handle, stream)`` entry of ``calls``; the few tensor
    ops in between (a fill, the exchange collective) are entries of their own.  Replaying the list re-runs the query
    without any planning / lowering / allocation / argument marshalling on the host.  A run that had to learn a size
    on the host mid-way (``poisoned``) is data-dependent and is not replayable."""

    def __init__(self) -> None:
        self.calls: list[tuple[Any, tuple]] = []
        self.keep: list[Any] = []
        self.poisoned = False
        self.finish: Any = None  # () -> (raw columns, nrows, flags)
        self.result: Any = None  # (schema, stage_id) of the writer
        self.self_cleaning = False  # the run's last launch resets the status words itself (no zeroing before a replay)
        self.segments: list[Any] = []  # capture handles owned by this recording
        self._free: Any = None

    def replay(self) -> bool:
        for fn, args in self.calls:
            rc = fn(*args)
            if rc:  # library calls return 0 on success; recorded tensor ops return None
                return False
        return True

    def __del__(self) -> None:
        if self._free is not None:
            for handle in self.segments:
                self._free(handle)
            self.segments = []


def _call_void(fn: Any, *args: Any) -> None:
    fn(*args)  # tensor ops return the tensor; a replayed call must report "no error"


class _RecordingLib:
    """Stands in for the ctypes library while a run is being recorded: the calls go through unchanged (the library
    itself captures their launches), their argument objects are kept alive with the recording."""

    def __init__(self, lib: Any, rec: Recording) -> None:
        self._lib, self._rec = lib, rec

    def __getattr__(self, name: str) -> Any:
        fn = getattr(self._lib, name)
        if fn.restype is not C.c_int or name.endswith("_geom") or name.endswith("_chunks"):
            return fn  # queries (sizes, error text) and host-side helpers are not device work

        def recorded(*args: Any) -> int:
            self._rec.keep.append(args)
            return fn(*args)

        return recorded


class Device:
    """One GPU + the loaded operator library."""

    def __init__(self, index: int = 0) -> None:
        self.lib = hs.load_library()
        if not torch.cuda.is_available():
            raise DeviceError(
                "HipExecutionEngine needs an AMD GPU (torch.cuda.is_available() is False); "
                "there is no CPU execution path in minispark_amd"
            )
        self.index = index
        self.device = torch.device("cuda", index)
        torch.cuda.set_device(self.device)
        self.flags = torch.zeros(4, dtype=torch.int32, device=self.device)
        self.last_global_tier = ""  # "radix" / "hash": what the last HBM-tier partial aggregate ran on
        self._partial_prepared: dict[Any, dict] = {}
        self._finish_prepared: dict[Any, dict] = {}
        self.zero_copy_results = os.environ.get("HIPSPARK_ZERO_COPY", "1") != "0"
        self._const_lens: dict[int, torch.Tensor] = {}
        self._raw_lib = self.lib
        self.rec: Recording | None = None
        self.scan_events = None
        self.exchange_events = None
        self.exchange_begins_at_scan_end = False  # the slab exchange's begin event is the timed scan launch's end event
        self.join_events = None

    def native_engine(self) -> Any:
        """The library's engine handle for this GPU (hs_engine_create): owner of the reader's pinned staging pool."""
        handle = self.__dict__.get("_native_engine")
        if handle is None:
            handle = C.c_void_p()
            hs.check(self._raw_lib.hs_engine_create(self.index, C.byref(handle)), "hs_engine_create")
            self.__dict__["_native_engine"] = handle
        return handle

    # ---- recording ------------------------------------------------------------------------------------
    def start_recording(self) -> Recording:
        self.rec = Recording()
        self.rec._free = self._raw_lib.hs_capture_free
        self.lib = _RecordingLib(self._raw_lib, self.rec)
        hs.check(self._raw_lib.hs_capture_begin(), "hs_capture_begin")
        return self.rec

    def _cut_segment(self, reopen: bool) -> None:
        """Close the library's open launch capture; the launches since the last cut become one replayable entry."""
        handle, n_ops = C.c_void_p(), C.c_int32(0)
        if self._raw_lib.hs_capture_end(C.byref(handle), C.byref(n_ops)) == 0 and handle.value:
            if n_ops.value > 0 and self.rec is not None:
                self.rec.segments.append(handle)
                self.rec.calls.append((self._raw_lib.hs_capture_replay, (handle, self.stream)))
            else:
                self._raw_lib.hs_capture_free(handle)
        if reopen:
            hs.check(self._raw_lib.hs_capture_begin(), "hs_capture_begin")

    def stop_recording(self) -> Recording | None:
        if self.rec is not None:
            self._cut_segment(reopen=False)
        rec, self.rec = self.rec, None
        self.lib = self._raw_lib
        return rec

    def op(self, fn: Any, *args: Any) -> None:
        """Run a small tensor op (fill / copy / collective) now and, when recording, on every replay."""
        fn(*args)
        if self.rec is not None:
            self._cut_segment(reopen=True)  # keep the order: library launches so far, then this op
            self.rec.calls.append((_call_void, (fn, *args)))

    def host_int(self, t: torch.Tensor) -> int:
        """A device value the HOST needs to go on (sizes an allocation): synchronises, and makes the current
        run data-dependent, i.e. not replayable."""
        if self.rec is not None:
            self.rec.poisoned = True
        return int(t.item())

    # ---- plumbing ------------------------------------------------------------------------------------
    def time_scan_kernel(self, enable: bool = True) -> None:
        """Bracket the main scan kernel of every following aggregate_partial with HIP events (recorded
        by the library on the launch stream); read with scan_kernel_ms() after a synchronise."""
        if enable:
            self.scan_events = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
            for ev in self.scan_events:
                ev.record()  # creates the underlying hipEvent_t
        else:
            self.scan_events = None

    def time_exchange(self, enable: bool = True) -> None:
        """Bracket the exchange collective of every following query with events on the launch stream (bench.py's
        time split); read with exchange_ms() after a synchronise."""
        if enable:
            # [0], [1]: around the query's first exchange (slabs / the join's build side); [2], [3]: around a second one
            # (the unit tables of the fused join)
            self.exchange_events = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            for ev in self.exchange_events:
                ev.record()
        else:
            self.exchange_events = None

    def exchange_ms(self) -> float:
        # a rank that does not receive the result returns from the query without waiting for its stream
        for ev in self.exchange_events:
            ev.synchronize()
        ev = self.exchange_events
        begin = self.scan_events[1] if self.exchange_begins_at_scan_end and self.scan_events is not None else ev[0]
        return begin.elapsed_time(ev[1]) + ev[2].elapsed_time(ev[3])

    def time_join(self, enable: bool = True) -> None:
        """Bracket the in-place join operator (table build + probe) of every following query with events on the
        launch stream - recorded as part of the run, so replayed runs are timed too; read with join_ms()."""
        if enable:
            self.join_events = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
            for ev in self.join_events:
                ev.record()
        else:
            self.join_events = None

    def join_ms(self) -> float:
        return self.join_events[0].elapsed_time(self.join_events[1])

    def _event_handle(self, i: int):
        ev = getattr(self, "scan_events", None)
        return None if ev is None else ev[i].cuda_event

    def scan_kernel_ms(self) -> float:
        self.scan_events[1].synchronize()  # see exchange_ms
        return self.scan_events[0].elapsed_time(self.scan_events[1])

    @property
    def stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def empty(self, n: int, dtype: torch.dtype) -> torch.Tensor:
        esize = torch.empty((), dtype=dtype).element_size()
        pad = (PAD + esize - 1) // esize
        t = torch.empty(int(n) + pad, dtype=dtype, device=self.device)[: int(n)]
        if self.rec is not None:
            self.rec.keep.append(t)
        return t

    def workspace(self, nbytes: int) -> torch.Tensor:
        t = torch.empty(max(int(nbytes), 256) + PAD, dtype=torch.uint8, device=self.device)
        if self.rec is not None:
            self.rec.keep.append(t)
        return t

    def to_device(self, arr: np.ndarray, dtype: torch.dtype | None = None) -> torch.Tensor:
        src = torch.from_numpy(np.require(arr, requirements=["C", "W"]))  # copies read-only views (np.frombuffer)
        dst = self.empty(src.numel(), dtype or src.dtype)
        dst.copy_(src, non_blocking=False)
        if self.rec is not None:
            self.rec.poisoned = True  # host data went into this run: re-running needs the host side again
        return dst

    def reset_flags(self) -> None:
        self.flags.zero_()
        for prep in self._partial_prepared.values():  # slab headers of a run that did not reach its finish launch
            if prep.get("tail") is not None:
                prep["slab"][:4].zero_()
        for prep in self._finish_prepared.values():  # ... and "done" words nobody read
            if prep["host_image"] is not None:
                prep["host_image"][4:8] = 0

    def read_flags(self) -> int:
        if self.rec is not None:
            self.rec.poisoned = True
        return int(self.flags[0].item()) & 0xFFFFFFFF

    def raise_for_flags(self, flags: int) -> None:
        """Data-dependent failures surface as the exception the reference's Python raises."""
        if flags & hs.FLAG_DIV_ZERO:
            raise ZeroDivisionError("division by zero")  # operator.truediv & co. in sql.py:262-266
        if flags & hs.FLAG_INT_OVERFLOW:
            raise OverflowError("int too big to convert")  # io.py:90
        if flags & hs.FLAG_FLT_OVERFLOW:
            raise OverflowError("float too large to pack with f format")  # io.py:94
        if flags & hs.FLAG_TYPE_ASSERT:
            # MIN/MAX identities are ints (constants.py:14-15): a FLOAT aggregate that never got below MAX_INT /
            # above MIN_INT is still an int when the reference writes it (tasks.py:303-310 -> io.py:93)
            raise AssertionError("FLOAT column holds int")
        if flags & hs.FLAG_STR_TOO_LONG:
            raise ValueError("string longer than 255 bytes cannot be stored in a BlockFile")
        if flags & hs.FLAG_BAD_PROGRAM:
            raise DeviceError("internal error: device interpreter rejected the program")
        if flags & hs.FLAG_PEER_TIMEOUT:
            raise DeviceError("peer-to-peer exchange: the partial rows of a rank did not arrive within the time limit")

    # ---- columns ---------------------------------------------------------------------------------
    def fixed_col(self, kind: int, arr: np.ndarray) -> DCol:
        t = self.to_device(arr, _TORCH_DTYPE[kind])
        return DCol(kind, t, int(t.numel()))

    def string_col(self, lens: torch.Tensor, data: torch.Tensor, n: int) -> DCol:
        """Finish a STRING column: offsets scan + fixed-length detection (one tiny D2H)."""
        offs = self.empty(n + 1, torch.int64)
        minmax = self.empty(2, torch.int32)
        ws = self.workspace(self.lib.hs_scan_ws_bytes(n))
        hs.check(self.lib.hs_str_offsets(self.stream, lens.data_ptr(), n, offs.data_ptr(), minmax.data_ptr(),
                                         ws.data_ptr()), "hs_str_offsets")
        if self.rec is not None:
            self.rec.poisoned = True
        mn, mx = minmax.tolist()
        fixed = mn if (n > 0 and mn == mx) else (0 if n == 0 else -1)
        return DCol(hs.STR, data, n, lens=lens, offs=None if fixed >= 0 else offs, fixed_len=fixed)

    def upload_raw(self, raw: Any, col_type: ColumnType) -> DCol:
        if col_type == ColumnType.STRING:
            lens = self.to_device(raw.lens, torch.uint8)
            data = self.to_device(raw.data, torch.uint8)
            return self.string_col(lens, data, len(raw))
        return self.fixed_col(FILE_KIND[col_type], raw)

    def download(self, col: DCol, col_type: ColumnType) -> Any:
        """Device column (already in file storage kinds) -> raw numpy column."""
        if col.kind == hs.STR:
            n = col.n
            if col.dict is not None:
                return _decode_codes(col.data[:n].cpu().numpy(), col.dict)
            lens = col.lens[:n].cpu().numpy()
            total = int(lens.sum(dtype=np.int64)) if col.fixed_len < 0 else n * col.fixed_len
            return StrCol(lens.astype(np.uint8), col.data[:total].cpu().numpy().astype(np.uint8))
        return col.data[: col.n].cpu().numpy()

    def download_batch(self, batch: DBatch, schema: Schema,
                       extra_flags: torch.Tensor | None = None) -> tuple[list[Any], int, int]:
        """All columns (already in file storage kinds) + the exact row count + the status flags in ONE
        device->host copy.  Returns (raw numpy columns, nrows, flags)."""
        parts: list[torch.Tensor] = []
        layout: list[tuple[str, int, int, Any]] = []
        pos = 0

        def add(tag: str, t: torch.Tensor, extra: Any = None) -> None:
            nonlocal pos
            b = t.contiguous().view(torch.uint8)
            parts.append(b)
            layout.append((tag, pos, b.numel(), extra))
            pos += b.numel()

        add("flags", self.flags[:1])
        if extra_flags is not None:
            add("xflags", extra_flags)
        if batch.nrows_dev is not None:
            add("nrows", batch.nrows_dev[:1])
        for c in batch.cols:
            if c.kind == hs.STR:
                add("lens", c.lens[: c.n])
                add("sdata", c.data, c)
            else:
                add("col", c.data[: c.n], c)
        def finish() -> tuple[list[Any], int, int]:
            host = torch.cat(parts).cpu().numpy()  # the single synchronising copy
            return self._parse_download(host, layout, batch.nrows)

        if self.rec is not None:
            self.rec.finish = finish
            self.rec.keep.append(parts)
        return finish()

    def _parse_download(self, host: np.ndarray, layout: list, nrows_max: int) -> tuple[list[Any], int, int]:
        flags = int(host[0:4].view(np.uint32)[0])
        batch_nrows = nrows_max
        n = batch_nrows
        for tag, off, size, _ in layout:
            if tag == "nrows":
                n = min(n, int(host[off: off + size].view(np.int64)[0]))
            elif tag == "xflags":  # status words of the other ranks (came with their slabs)
                for w in host[off: off + size].view(np.uint32):
                    flags |= int(w)
        raw: list[Any] = []
        pending_lens = None
        np_dtype = {hs.I32: np.int32, hs.F32: np.float32, hs.I64: np.int64, hs.F64: np.float64, hs.U8: np.uint8}
        for tag, off, size, c in layout:
            if tag == "lens":
                pending_lens = host[off: off + size][:n].copy()
            elif tag == "sdata":
                total = int(pending_lens.sum(dtype=np.int64))
                if c.dict is not None:  # one code byte per row -> the strings they stand for
                    raw.append(_decode_codes(host[off: off + size][:n], c.dict))
                else:
                    raw.append(StrCol(pending_lens, host[off: off + size][:total].copy()))
            elif tag == "col":
                raw.append(host[off: off + size].view(np_dtype[c.kind])[:n].copy())
        return raw, n, flags

    # ---- expression evaluation (A4) ---------------------------------------------------------------------
    def _cols_array(self, batch: DBatch, order: Sequence[int], code_columns: Sequence[int] = ()):
        arr = (hs.hs_col * max(len(order), 1))()
        for s, idx in enumerate(order):
            arr[s] = batch.cols[idx].as_hs()
            if idx in code_columns:  # only its code bytes are read (HS_OP_DICTBIT): a plain byte column to the kernels
                arr[s].kind, arr[s].fixed_len = hs.U8, -1
        return arr

    def eval_numeric(self, batch: DBatch, exprs: Sequence[Any], sel: torch.Tensor | None = None,
                     n: int | None = None) -> list[tuple[DCol, str]]:
        """Evaluate numeric / boolean expressions -> in-flight columns (F64 / I64 / U8 mask)."""
        n = batch.nrows if n is None else n
        while True:
            b = ProgramBuilder(batch.schema, batch.kinds, batch.dicts)
            try:
                tags = [b.emit_out(i, e) for i, e in enumerate(exprs)]
                break
            except NeedsDecoded as e:  # a coded column used in a way that has no coded form
                batch = self.decoded_batch(batch, [e.column])
        prog = b.finish()
        out_kinds = [hs.F64 if t == "F" else (hs.U8 if t == "B" else hs.I64) for t in tags]
        outs = [self.empty(n, _TORCH_DTYPE[k]) for k in out_kinds]
        if n > 0:
            cols = self._cols_array(batch, prog.columns, prog.code_columns)
            out_ptrs = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
            kinds = (C.c_int32 * len(outs))(*out_kinds)
            pstruct = prog.to_struct()
            hs.check(self.lib.hs_eval(self.stream, cols, len(prog.columns), C.byref(pstruct),
                                      sel.data_ptr() if sel is not None else None, n,
                                      batch.n_dev_ptr if sel is None else None, out_ptrs, kinds, len(outs),
                                      self.flags.data_ptr()), "hs_eval")
        return [(DCol(k, o, n), t) for k, o, t in zip(out_kinds, outs, tags)]

    # ---- filter (A3) -----------------------------------------------------------------------------------
    def filter_select(self, batch: DBatch, conds: Sequence[Any]) -> tuple[torch.Tensor, int]:
        """WHERE -> ascending indices of the surviving rows (+ their number; one D2H)."""
        batch = self.resolve(batch)
        n = batch.nrows
        cond = conds[0]
        for extra in conds[1:]:
            cond = cond & extra
        (mask, tag), = self.eval_numeric(batch, [cond])
        if tag not in ("B",):
            (mask, _), = self.eval_numeric(batch, [cond != 0])
        sel = self.empty(n, torch.int64)
        count = self.empty(1, torch.int64)
        ws = self.workspace(self.lib.hs_scan_ws_bytes(n))
        hs.check(self.lib.hs_compact(self.stream, mask.data.data_ptr(), n, sel.data_ptr(), count.data_ptr(),
                                     ws.data_ptr()), "hs_compact")
        return sel, self.host_int(count)

    # ---- gathers -------------------------------------------------------------------------------------
    def gather_col(self, col: DCol, idx: torch.Tensor, n: int, n_dev: torch.Tensor | None = None,
                   out: torch.Tensor | None = None) -> DCol:
        """out[i] = col[idx[i]], i < n (and < n_dev[0] when given).  Variable-length strings need the
        exact count (their payload size comes from a scan), fixed-length ones do not."""
        n_dev_ptr = n_dev.data_ptr() if n_dev is not None else None
        if col.kind == hs.STR and col.fixed_len in (1, 2, 4, 8):
            width = col.fixed_len
            data = out if out is not None else self.empty(n * width, torch.uint8)
            hs.check(self.lib.hs_gather_fixed(self.stream, col.data.data_ptr(), width, col.n, idx.data_ptr(), n, n_dev_ptr,
                                              data.data_ptr(), self.flags.data_ptr()), "hs_gather_fixed")
            return DCol(hs.STR, data, n, lens=self.const_lens(width, n), offs=None, fixed_len=width, dict=col.dict)
        if col.kind == hs.STR:
            if n_dev is not None:
                n = min(n, self.host_int(n_dev[0]))
            if self.rec is not None:
                self.rec.poisoned = True  # payload size of variable-length strings is learnt on the host
            src = col.as_hs()
            lens = self.empty(n, torch.uint8)
            hs.check(self.lib.hs_gather_str_lens(self.stream, C.byref(src), col.n, idx.data_ptr(), n, lens.data_ptr(),
                                                 self.flags.data_ptr()), "hs_gather_str_lens")
            offs = self.empty(n + 1, torch.int64)
            minmax = self.empty(2, torch.int32)
            ws = self.workspace(self.lib.hs_scan_ws_bytes(n))
            hs.check(self.lib.hs_str_offsets(self.stream, lens.data_ptr(), n, offs.data_ptr(), minmax.data_ptr(),
                                             ws.data_ptr()), "hs_str_offsets")
            total = int(offs[n].item()) if n > 0 else 0
            mn, mx = minmax.tolist()
            data = self.empty(total, torch.uint8)
            hs.check(self.lib.hs_gather_str_bytes(self.stream, C.byref(src), col.n, idx.data_ptr(), n, offs.data_ptr(),
                                                  data.data_ptr()), "hs_gather_str_bytes")
            fixed = mn if (n > 0 and mn == mx) else (0 if n == 0 else -1)
            return DCol(hs.STR, data, n, lens=lens, offs=None if fixed >= 0 else offs, fixed_len=fixed)
        if col.kind == hs.STR and out is not None:
            raise SlabUnsupported("variable-length string keys cannot be written into a fixed-size exchange slab")
        out = out if out is not None else self.empty(n, _TORCH_DTYPE[col.kind])
        hs.check(self.lib.hs_gather_fixed(self.stream, col.data.data_ptr(), hs.KIND_BYTES[col.kind], col.n, idx.data_ptr(),
                                          n, n_dev_ptr, out.data_ptr(), self.flags.data_ptr()), "hs_gather_fixed")
        return DCol(col.kind, out, n)

    def const_lens(self, width: int, n: int) -> torch.Tensor:
        """Length bytes of a fixed-width string column (all equal): one shared read-only buffer per width."""
        buf = self._const_lens.get(width)
        if buf is None or buf.numel() < n:
            buf = torch.full((max(n, 1024) + PAD,), width, dtype=torch.uint8, device=self.device)
            self._const_lens[width] = buf
        return buf[:n]

    def with_string_width(self, col: DCol, width: int) -> DCol:
        """The same STRING column described with the width agreed over all ranks (distributed.agree_string_width):
        an empty local column adopts the peers' fixed width; a locally fixed-width column whose peers hold other
        lengths gets its offsets materialised (``fixed_len`` -1) so every rank runs the same code path."""
        if col.kind != hs.STR or col.fixed_len == width:
            return col
        if width >= 0:
            if col.n != 0:
                raise DeviceError(f"ranks disagree: local strings are not {width} bytes long")
            return DCol(hs.STR, col.data, 0, lens=self.const_lens(width, 0), offs=None, fixed_len=width)
        n = col.n
        lens = col.lens if col.lens is not None else self.const_lens(max(col.fixed_len, 0), n)
        offs = self.empty(n + 1, torch.int64)
        minmax = self.empty(2, torch.int32)
        ws = self.workspace(self.lib.hs_scan_ws_bytes(n))
        hs.check(self.lib.hs_str_offsets(self.stream, lens.data_ptr(), n, offs.data_ptr(), minmax.data_ptr(),
                                         ws.data_ptr()), "hs_str_offsets")
        return DCol(hs.STR, col.data, n, lens=lens, offs=offs, fixed_len=-1)

    def resolve(self, batch: DBatch) -> DBatch:
        """Make a lazily-sized batch exact (one D2H of the row count)."""
        if not batch.lazy:
            return batch
        n = min(batch.nrows, self.host_int(batch.nrows_dev[0]))
        cols = []
        for c in batch.cols:
            if c.kind == hs.STR:
                cols.append(DCol(hs.STR, c.data, n, lens=c.lens[:n] if c.lens is not None else None,
                                 offs=c.offs[: n + 1] if c.offs is not None else None, fixed_len=c.fixed_len,
                                 dict=c.dict))
            else:
                cols.append(DCol(c.kind, c.data[:n], n))
        return DBatch(list(batch.schema), cols, n, [0, n])

    def gather_batch(self, batch: DBatch, idx: torch.Tensor, n: int, unit_rows: list[int] | None = None) -> DBatch:
        return DBatch(list(batch.schema), [self.gather_col(c, idx, n) for c in batch.cols], n,
                      unit_rows or [0, n])

    # ---- dictionary-coded strings ---------------------------------------------------------------------------
    DICT_SLOTS = 4096  # slots of the string set built on the device (a column with <= 256 distinct strings is coded)

    def dict_encode(self, col: DCol) -> DCol | None:
        """`_dict_encode` with its wall time added to ``dict_encode_seconds`` (it ends on a read-back, so the clock sees the
        kernels too): the table-open share of a cold query (bench.py `cold`)."""
        t0 = time.perf_counter()
        try:
            return self._dict_encode(col)
        finally:
            self.dict_encode_seconds = getattr(self, "dict_encode_seconds", 0.0) + time.perf_counter() - t0

    def _dict_encode(self, col: DCol) -> DCol | None:
        """Try to re-code a STRING column as one byte per row + dictionary (at most 256 distinct strings; sorted, so
        the codes do not depend on row order).  None = the column stays as it is.  Two passes over the column
        (hs_dict_build / hs_dict_assign) and one small read-back: done once, when a table column is loaded."""
        if col.kind != hs.STR or col.dict is not None or col.n == 0:
            return None
        cap = self.DICT_SLOTS
        words = self.empty(cap, torch.int64)
        reps = self.empty(cap, torch.int64)
        state = torch.zeros(2, dtype=torch.int32, device=self.device)  # [count, flags]
        src = col.as_hs()
        hs.check(self._raw_lib.hs_dict_build(self.stream, C.byref(src), col.n, cap, words.data_ptr(), reps.data_ptr(),
                                             state[0:].data_ptr(), state[1:].data_ptr()), "hs_dict_build")
        count, flags = (int(v) for v in state.tolist())
        if flags or count == 0:
            return None
        slot_reps = reps.cpu().numpy()
        slots = np.nonzero(slot_reps >= 0)[0]
        rep_rows = torch.from_numpy(slot_reps[slots].astype(np.int64)).to(self.device)
        texts = _strcol_bytes(self.download(self.gather_col(col, rep_rows, len(slots)), ColumnType.STRING))
        # codes are named by STRING: a string that landed in several slots (hs_dict_build, contention) has one code
        entries = sorted(set(texts))
        if len(entries) > 256:
            return None
        code_of = {text: code for code, text in enumerate(entries)}
        slot_code = np.zeros(cap, dtype=np.uint8)
        for slot, text in zip(slots.tolist(), texts):
            slot_code[slot] = code_of[text]
        codes = self.empty(col.n, torch.uint8)
        d_slot_code = torch.from_numpy(slot_code).to(self.device)
        hs.check(self._raw_lib.hs_dict_assign(self.stream, C.byref(src), col.n, cap, words.data_ptr(), reps.data_ptr(),
                                              d_slot_code.data_ptr(), codes.data_ptr(), state[1:].data_ptr()),
                 "hs_dict_assign")
        if int(state[1].item()):
            return None
        return DCol(hs.STR, codes, col.n, lens=self.const_lens(1, col.n), offs=None, fixed_len=1,
                    dict=tuple(entries), plain=col)

    def dict_recode(self, plain: DCol, coded: DCol | None, entries: tuple) -> DCol:
        """The coded column re-expressed in another (larger) dictionary `entries` that holds every string of its own:
        code bytes through a 256-entry table (hs_remap_u8).  `coded` None = a column without rows."""
        n = plain.n
        out = self.empty(n, torch.uint8)
        if coded is not None and n > 0:
            index = {text: code for code, text in enumerate(entries)}
            lut = np.zeros(256, dtype=np.uint8)
            for old, text in enumerate(coded.dict):
                lut[old] = index[text]
            d_lut = torch.from_numpy(lut).to(self.device)
            hs.check(self._raw_lib.hs_remap_u8(self.stream, coded.data.data_ptr(), n, d_lut.data_ptr(), out.data_ptr()),
                     "hs_remap_u8")
            torch.cuda.current_stream(self.device).synchronize()  # d_lut goes out of scope: table open, not a query
        return DCol(hs.STR, out, n, lens=self.const_lens(1, n), offs=None, fixed_len=1, dict=tuple(entries), plain=plain)

    def dict_column(self, entries: tuple) -> DCol:
        """The dictionary's strings as a device STRING column (row = code): the source of decoding gathers."""
        cache = self.__dict__.setdefault("_dict_columns", {})
        hit = cache.get(entries)
        if hit is None:
            lens = np.array([len(e) for e in entries], dtype=np.uint8)
            data = np.frombuffer(b"".join(entries) or b"\0", dtype=np.uint8)
            recording, self.rec = self.rec, None  # a constant of the query shape, not host data of this run
            try:
                hit = self.string_col(self.to_device(lens, torch.uint8), self.to_device(data, torch.uint8), len(entries))
            finally:
                self.rec = recording
            if len(cache) >= 64:
                cache.pop(next(iter(cache)))
            cache[entries] = hit
        return hit

    def decoded(self, col: DCol) -> DCol:
        """A dictionary-coded column as real strings (lens + bytes [+ offsets]); any other column unchanged."""
        if col.dict is None:
            return col
        if col.plain is not None and col.plain.n == col.n:
            return col.plain
        idx = self.empty(col.n, torch.int64)
        idx.copy_(col.data[: col.n])  # code byte -> row of the dictionary column (widening copy: plumbing)
        return self.gather_col(self.dict_column(col.dict), idx, col.n)

    def decoded_batch(self, batch: DBatch, which: Sequence[int] | None = None) -> DBatch:
        """The batch with its dictionary-coded columns (all, or the listed ones) decoded."""
        import dataclasses  # noqa: PLC0415

        todo = [i for i, c in enumerate(batch.cols) if c.dict is not None and (which is None or i in which)]
        if not todo:
            return batch
        cols = list(batch.cols)
        for i in todo:
            cols[i] = self.decoded(cols[i])
        return dataclasses.replace(batch, cols=cols)

    def dict_concat(self, batch: DBatch, parts: StringParts, n: int) -> DCol | None:
        """Concatenation whose column parts are all dictionary-coded: the result is coded too - codes combined in
        mixed radix (one launch over code bytes), dictionary = the product of the parts' dictionaries with the
        literals spliced in.  None when that product would not fit a code byte."""
        col_parts = [(i, v) for i, (what, v) in enumerate(parts.parts) if what == "col"]
        if not col_parts or any(batch.cols[v].dict is None for _, v in col_parts) or len(col_parts) > 4:
            return None
        sizes = [len(batch.cols[v].dict) for _, v in col_parts]
        total = 1
        for sz in sizes:
            total *= sz
        if total > 256:
            return None
        strides, acc = [], 1
        for sz in reversed(sizes):  # the last column part varies fastest
            strides.insert(0, acc)
            acc *= sz
        entries = []
        for code in range(total):
            text, k = b"", 0
            for what, v in parts.parts:
                if what == "col":
                    text += batch.cols[v].dict[(code // strides[k]) % sizes[k]]
                    k += 1
                else:
                    text += v
            if len(text) > 255:
                return None  # the reference fails at the file write; the generic path reports it per row
            entries.append(text)
        if len(set(entries)) != len(entries):
            # two code combinations spell the same string ({'a','ab'} + {'bc','c'}: 'a'+'bc' == 'ab'+'c'): a GROUP BY
            # on the code byte would return that string twice; the reference groups on the string.  Plain strings then.
            return None
        if any(batch.cols[v].data.data_ptr() % 16 for _, v in col_parts):
            return None  # the combine reads 16 code bytes per lane
        out = self.empty(n, torch.uint8)
        if n > 0:  # an empty batch (every row filtered out) has no buffers to hand over
            ptrs = (C.c_void_p * len(col_parts))(*[batch.cols[v].data.data_ptr() for _, v in col_parts])
            strd = (C.c_int32 * len(col_parts))(*strides)
            hs.check(self.lib.hs_dict_combine(self.stream, len(col_parts), ptrs, strd, n, out.data_ptr()), "hs_dict_combine")
        return DCol(hs.STR, out, n, lens=self.const_lens(1, n), offs=None, fixed_len=1, dict=tuple(entries))

    # ---- string concat -------------------------------------------------------------------------------
    def concat_strings(self, batch: DBatch, parts: StringParts, n: int) -> DCol:
        coded = self.dict_concat(batch, parts, n)
        if coded is not None:
            return coded
        batch = self.decoded_batch(batch, [v for what, v in parts.parts if what == "col"])
        if len(parts.parts) > hs.HS_MAX_PARTS:
            raise NotImplementedError(f"string concatenation of more than {hs.HS_MAX_PARTS} parts")
        arr = (hs.hs_col * len(parts.parts))()
        keep = []
        for i, (what, value) in enumerate(parts.parts):
            if what == "col":
                arr[i] = batch.cols[value].as_hs()
            else:
                lit = self.to_device(np.frombuffer(value or b"\0", dtype=np.uint8), torch.uint8)
                keep.append(lit)
                c = hs.hs_col()
                c.kind = -1
                c.fixed_len = len(value)
                c.data = lit.data_ptr()
                arr[i] = c
        lens = self.empty(n, torch.uint8)
        hs.check(self.lib.hs_concat_lens(self.stream, arr, len(parts.parts), n, lens.data_ptr(), self.flags.data_ptr()),
                 "hs_concat_lens")
        offs = self.empty(n + 1, torch.int64)
        minmax = self.empty(2, torch.int32)
        ws = self.workspace(self.lib.hs_scan_ws_bytes(n))
        hs.check(self.lib.hs_str_offsets(self.stream, lens.data_ptr(), n, offs.data_ptr(), minmax.data_ptr(),
                                         ws.data_ptr()), "hs_str_offsets")
        total = int(offs[n].item()) if n > 0 else 0
        mn, mx = minmax.tolist()
        data = self.empty(total, torch.uint8)
        hs.check(self.lib.hs_concat_bytes(self.stream, arr, len(parts.parts), n, offs.data_ptr(), data.data_ptr()),
                 "hs_concat_bytes")
        fixed = mn if (n > 0 and mn == mx) else (0 if n == 0 else -1)
        return DCol(hs.STR, data, n, lens=lens, offs=None if fixed >= 0 else offs, fixed_len=fixed)

    # ---- quantisation (A6 / K12) -------------------------------------------------------------------------
    def quantise_col(self, col: DCol, col_type: ColumnType, n_dev_ptr=None) -> DCol:
        """In-flight column -> the storage kind a BlockFile holds (f64->f32, i64->i32)."""
        want = FILE_KIND[col_type]
        if col.kind == want or col.kind == hs.STR:
            return col
        if col.kind == hs.U8:
            raise AssertionError("a boolean column cannot be written to a BlockFile")  # reference: io.py:89 assert
        if (col.kind, want) in ((hs.F64, hs.F32), (hs.I64, hs.I32)):
            out = self.empty(col.n, _TORCH_DTYPE[want])
            hs.check(self.lib.hs_quantise(self.stream, col.data.data_ptr(), col.kind, col.n, n_dev_ptr, out.data_ptr(),
                                          self.flags.data_ptr()), "hs_quantise")
            return DCol(want, out, col.n)
        if (col.kind, want) == (hs.I64, hs.I64):
            return col
        raise AssertionError(f"column of kind {col.kind} cannot be stored as {col_type}")

    def quantise_cols(self, cols: Sequence[DCol], types: Sequence[ColumnType], n_dev_ptr=None) -> list[DCol]:
        """quantise_col for a whole batch with ONE launch for all the columns that need converting."""
        out: list[DCol] = list(cols)
        todo = []
        for i, (c, t) in enumerate(zip(cols, types)):
            want = FILE_KIND[t]
            if c.kind == want or c.kind == hs.STR or (c.kind, want) == (hs.I64, hs.I64):
                continue
            if (c.kind, want) not in ((hs.F64, hs.F32), (hs.I64, hs.I32)):
                out[i] = self.quantise_col(c, t, n_dev_ptr)  # raises the appropriate error
                continue
            todo.append((i, c, want))
        for lo in range(0, len(todo), 16):
            part = todo[lo: lo + 16]
            n = part[0][1].n
            if any(c.n != n for _, c, _ in part):
                for i, c, _ in part:
                    out[i] = self.quantise_col(c, types[i], n_dev_ptr)
                continue
            dsts = [self.empty(n, _TORCH_DTYPE[w]) for _, _, w in part]
            srcs_arr = (C.c_void_p * len(part))(*[c.data.data_ptr() for _, c, _ in part])
            kinds_arr = (C.c_int32 * len(part))(*[c.kind for _, c, _ in part])
            dsts_arr = (C.c_void_p * len(part))(*[d.data_ptr() for d in dsts])
            hs.check(self.lib.hs_quantise_many(self.stream, len(part), srcs_arr, kinds_arr, n, n_dev_ptr, dsts_arr,
                                               self.flags.data_ptr()), "hs_quantise_many")
            for (i, _, w), d in zip(part, dsts):
                out[i] = DCol(w, d, n)
        return out

    # ---- partial aggregate (A5/A6) -------------------------------------------------------------------------
    def aggregate_partial(self, batch: DBatch, filters: Sequence[Any], group_by: Any, agg_columns: Sequence[Any],
                          out_schema: Schema, group_cap_hint: int = 4, cache_key: Any = None,
                          slab_rows: int | None = None, tail: bool = False, shared: bool = False) -> DBatch:
        """_aggregate_partial over a batch whose dictionary-coded columns are decoded where the query uses them in a
        way that has no coded form (remembered per query node, so later runs decode up front)."""
        hints = self.__dict__.setdefault("_decode_hints", {})
        todo = hints.get(cache_key) if cache_key is not None else None
        if todo:
            batch = self.decoded_batch(batch, todo)
        while True:
            try:
                return self._aggregate_partial(batch, filters, group_by, agg_columns, out_schema, group_cap_hint,
                                               cache_key, slab_rows, tail, shared)
            except NeedsDecoded as e:
                batch = self.decoded_batch(batch, [e.column])
                if cache_key is not None:
                    if len(hints) >= 64:
                        hints.pop(next(iter(hints)))
                    hints.setdefault(cache_key, set()).add(e.column)

    def _aggregate_partial(self, batch: DBatch, filters: Sequence[Any], group_by: Any, agg_columns: Sequence[Any],
                           out_schema: Schema, group_cap_hint: int = 4, cache_key: Any = None,
                           slab_rows: int | None = None, tail: bool = False, shared: bool = False) -> DBatch:
        """Fused scan + WHERE + aggregate arguments + per-unit partial aggregate.

        Returns the partial rows exactly as the reference would have written them to its shuffle
        file: key column + one column per aggregate, FLOAT partials rounded to f32, INTEGER partials
        range-checked to i32, rows grouped by unit.  Everything that depends only on (query node, input
        buffers, capacity) - lowered program, geometry, unit tables, output buffers - is prepared once
        and re-used by later runs of the same query (``cache_key``)."""
        batch = self.resolve(batch)  # units are row ranges: the row count must be exact
        if batch.nrows == 0:  # e.g. a rank that owns no block of a small table
            return self._empty_partial(batch, filters, group_by, agg_columns, out_schema, slab_rows, tail)
        cap = max(1, int(group_cap_hint))
        key = None
        if cache_key is not None:
            # every pointer the prepared column table holds: a computed string column is rebuilt per run and the
            # allocator may hand back the same data block with different lens / offs blocks
            key = (cache_key, cap, batch.nrows,
                   tuple((c.kind, c.fixed_len, c.data.data_ptr(), c.n, c.lens.data_ptr() if c.lens is not None else 0,
                          c.offs.data_ptr() if c.offs is not None else 0) for c in batch.cols),
                   tuple(batch.unit_rows), tuple(batch.unit_ids) if batch.unit_ids is not None else None, slab_rows,
                   tail, shared, batch.unit_col.data_ptr() if batch.unit_col is not None else 0, batch.n_unit_ids)
        prep = self._partial_prepared.get(key) if key is not None else None
        if prep is None:
            prep = self._prepare_partial(batch, filters, group_by, agg_columns, cap, slab_rows, tail, out_schema, shared)
            if key is not None:
                if len(self._partial_prepared) >= 8:
                    self._partial_prepared.pop(next(iter(self._partial_prepared)))
                self._partial_prepared[key] = prep
        p = prep
        if self.rec is not None:
            # a recording replays raw pointers into this entry's buffers: it must outlive the cache's eviction
            self.rec.keep.append(p)
        if tail:
            # the unit combine writes the partial rows (stored kinds) straight into the slab: no pack, no gather
            hs.check(self.lib.hs_agg_partial_slab(
                self.stream, p["cols"], p["n_cols"], p["key_slot"], C.byref(p["prog"]), C.byref(p["spec"]),
                p["d_units"].data_ptr(), p["d_chunk0"].data_ptr(), p["n_units"], C.byref(p["geom"]),
                p["d_unit_ids"].data_ptr() if p["d_unit_ids"] is not None else None, p["slab"].data_ptr(),
                C.byref(p["desc"]), p["ws"].data_ptr(), self.flags.data_ptr(), self._event_handle(0),
                self._event_handle(1)), "hs_agg_partial_slab")
            self.last_scan = p["info"]
            self.last_group_cap = cap
            return DBatch(list(out_schema), [], p["slots"], [0, p["slots"]], None, total_units=batch.total_units,
                          slab=p["slab"], slab_layout=p["layout"], tail=p["tail"])
        if shared and p.get("unit_slot") is not None:
            # computed units (probe side of a join): the unit id column rides along as one more preloaded slot
            hs.check(self.lib.hs_agg_shared_units(self.stream, p["cols"], p["n_cols"], p["key_slot"], p["unit_slot"],
                                                  p["n_units"], C.byref(p["prog"]), C.byref(p["spec"]),
                                                  p["d_units"].data_ptr(), C.byref(p["geom"]), p["out_rep"].data_ptr(),
                                                  p["out_acc"].data_ptr(), p["ngroups"].data_ptr(), p["ws"].data_ptr(),
                                                  self.flags.data_ptr(), self._event_handle(0), self._event_handle(1)),
                     "hs_agg_shared_units")
        elif shared:
            # tens to thousands of groups per unit: one LDS table per workgroup, LDS atomics
            hs.check(self.lib.hs_agg_shared(self.stream, p["cols"], p["n_cols"], p["key_slot"], C.byref(p["prog"]),
                                            C.byref(p["spec"]), p["d_units"].data_ptr(), p["n_units"], C.byref(p["geom"]),
                                            p["out_rep"].data_ptr(), p["out_acc"].data_ptr(), p["ngroups"].data_ptr(),
                                            p["ws"].data_ptr(), self.flags.data_ptr(), self._event_handle(0),
                                            self._event_handle(1)), "hs_agg_shared")
        else:
            hs.check(self.lib.hs_agg_partial(self.stream, p["cols"], p["n_cols"], p["key_slot"], C.byref(p["prog"]),
                                             C.byref(p["spec"]), p["d_units"].data_ptr(), p["d_chunk0"].data_ptr(),
                                             p["n_units"], C.byref(p["geom"]), p["out_rep"].data_ptr(),
                                             p["out_acc"].data_ptr(), p["ngroups"].data_ptr(), p["ws"].data_ptr(),
                                             self.flags.data_ptr(), self._event_handle(0), self._event_handle(1)),
                     "hs_agg_partial")
        self.last_scan = p["info"]
        hs.check(self.lib.hs_agg_pack(self.stream, p["out_rep"].data_ptr(), p["out_acc"].data_ptr(),
                                      p["ngroups"].data_ptr(), p["n_units"], p["unit_cap"], C.byref(p["spec"]),
                                      p["pack_start"].data_ptr(), p["dense_rep"].data_ptr(), p["out_ptrs"],
                                      p["kinds_arr"], p["d_unit_ids"].data_ptr() if p["d_unit_ids"] is not None else None,
                                      p["out_unit"].data_ptr() if p["out_unit"] is not None else None), "hs_agg_pack")
        # No host round trip here: the number of partial rows stays on the device (pack_start[n_units]);
        # a dictionary overflow is noticed at the query's final read-back and the query re-run.
        n_max = p["slots"]
        n_dev = p["pack_start"][p["n_units"]:]
        key_col = self.gather_col(batch.cols[p["key_idx"]], p["dense_rep"], n_max, n_dev, out=p["key_out"])
        if key_col.n != n_max:  # variable-length string keys made the count exact
            n_max, n_dev = key_col.n, None
        out_cols = [key_col]
        for acc in p["agg_to_acc"]:
            out_cols.append(DCol(p["acc_kinds"][acc], p["acc_bufs"][acc][:n_max], n_max))
        self.last_group_cap = cap
        if p["slab"] is not None:  # header of the exchange slab: status so far + number of rows
            self.op(p["layout"].flags_view(p["slab"]).copy_, self.flags[:1])
            self.op(p["layout"].count_view(p["slab"]).copy_, p["pack_start"][p["n_units"]: p["n_units"] + 1])
        return DBatch(list(out_schema), out_cols, n_max, None, n_dev, order=p["out_unit"], slab=p["slab"],
                      slab_layout=p["layout"], total_units=batch.total_units,
                      slab_cols=[0] + [1 + acc for acc in p["agg_to_acc"]])

    def _empty_partial(self, batch: DBatch, filters: Sequence[Any], group_by: Any, agg_columns: Sequence[Any],
                       out_schema: Schema, slab_rows: int | None, tail: bool = False) -> DBatch:
        """Partial aggregate of zero rows (a rank that owns no block): no launch, but the same column /
        slab layout as the other ranks so the exchange stays symmetric."""
        low = lower_aggregate(batch.schema, batch.kinds, filters, group_by, agg_columns, batch.dicts)
        key_src = batch.cols[low.program.columns[low.key_slot]]
        acc_kinds = [hs.I32 if is_int else hs.F32 for is_int in low.acc_is_int]
        if tail:
            if slab_rows is None:
                raise SlabUnsupported("an empty single-GPU input takes the general path")
            layout, slab, desc = self._tail_slab(key_src, out_schema, acc_kinds, slab_rows)
            info = {"desc": desc, "layout": layout, "slab": slab, "agg_to_acc": list(low.agg_to_acc),
                    "acc_kinds": acc_kinds, "key_kind": key_src.kind, "key_len": key_src.fixed_len, "n_units": 0,
                    "key_dict": key_src.dict}
            return DBatch(list(out_schema), [], 0, [0, 0], None, total_units=batch.total_units, slab=slab,
                          slab_layout=layout, tail=info)
        if slab_rows is not None and key_src.kind == hs.STR and key_src.fixed_len not in (1, 2, 4, 8):
            raise SlabUnsupported("variable-length string GROUP BY key")
        cols = [self.gather_col(key_src, self.empty(0, torch.int64), 0)]
        cols += [DCol(acc_kinds[acc], self.empty(0, _TORCH_DTYPE[acc_kinds[acc]]), 0) for acc in low.agg_to_acc]
        out = DBatch(list(out_schema), cols, 0, [0, 0], order=self.empty(0, torch.int64), total_units=batch.total_units)
        if slab_rows is not None:
            from .distributed import SlabLayout  # noqa: PLC0415

            key_spec = ((key_src.fixed_len, torch.uint8) if key_src.kind == hs.STR
                        else (hs.KIND_BYTES[key_src.kind], _TORCH_DTYPE[key_src.kind]))
            out.slab_layout = SlabLayout.build(slab_rows, [key_spec] + [(4, _TORCH_DTYPE[k]) for k in acc_kinds])
            out.slab = torch.zeros(out.slab_layout.nbytes, dtype=torch.uint8, device=self.device)  # count = 0
            out.slab_cols = [0] + [1 + acc for acc in low.agg_to_acc]
        return out

    def _tail_slab(self, key_col: DCol, out_schema: Schema, acc_kinds: Sequence[int], rows: int):
        """Slab (order keys -1 = no rows yet) + its description for the short tail.  Keys must be stored
        kinds that pack into the 64-bit key word."""
        from .distributed import SlabLayout  # noqa: PLC0415

        if key_col.kind == hs.STR:
            if key_col.fixed_len not in (1, 2, 4):
                raise SlabUnsupported("string GROUP BY key without a short fixed length")
            key_spec = (key_col.fixed_len, torch.uint8)
        else:
            if key_col.kind not in (hs.I32, hs.F32, hs.I64) or key_col.kind != FILE_KIND[out_schema[0][1]]:
                raise SlabUnsupported("GROUP BY key is not in its stored kind")
            key_spec = (hs.KIND_BYTES[key_col.kind], _TORCH_DTYPE[key_col.kind])
        layout = SlabLayout.build(rows, [key_spec] + [(4, _TORCH_DTYPE[k]) for k in acc_kinds])
        slab = torch.zeros(layout.nbytes, dtype=torch.uint8, device=self.device)
        layout.order_view(slab).fill_(-1)
        desc = hs.hs_slab_desc()
        desc.slab_rows, desc.stride, desc.order_off = rows, layout.nbytes, layout.order_offset
        desc.key_off = layout.columns[0].offset
        desc.key_kind, desc.key_len = key_col.kind, max(key_col.fixed_len, 0)
        desc.n_acc = len(acc_kinds)
        for a, k in enumerate(acc_kinds):
            desc.acc_off[a] = layout.columns[1 + a].offset
            desc.acc_kind[a] = k
        return layout, slab, desc

    def _prepare_partial(self, batch: DBatch, filters: Sequence[Any], group_by: Any, agg_columns: Sequence[Any],
                         cap: int, slab_rows: int | None = None, tail: bool = False,
                         out_schema: Schema | None = None, shared: bool = False) -> dict:
        low = lower_aggregate(batch.schema, batch.kinds, filters, group_by, agg_columns, batch.dicts)
        if low.numeric_slots > hs.HS_FUSED_COLS:
            raise TierExceeded(f"aggregate reads more than {hs.HS_FUSED_COLS} numeric columns")
        n_units = batch.n_units
        n_acc = len(low.acc_ops)
        computed = batch.unit_col is not None
        if computed:
            # units are per-row ids: ONE row range for the chunking, batch.n_unit_ids unit tables for the outputs
            if not shared or tail or slab_rows is not None:
                raise TierExceeded("computed units run on the shared-dictionary tier only")
            if len(low.program.columns) >= hs.HS_FUSED_COLS:
                raise TierExceeded("no preloaded column slot left for the unit ids")
            kc = batch.cols[low.program.columns[low.key_slot]]
            if not (kc.kind in (hs.I32, hs.U8) or (kc.kind == hs.STR and 1 <= kc.fixed_len <= 6)):
                raise TierExceeded("computed units need a GROUP BY key of at most 56 bits")
            host_units = (C.c_int64 * 2)(0, batch.nrows)
            n_units = 1
        else:
            host_units = (C.c_int64 * (n_units + 1))(*batch.unit_rows)
        geom = hs.hs_agg_geom()
        if shared:
            rc = self.lib.hs_agg_shared_geom(host_units, n_units, n_acc, cap, C.byref(geom))
        else:
            rc = self.lib.hs_agg_partial_geom(host_units, n_units, n_acc, cap, C.byref(geom))
        if rc == 2:
            raise TierExceeded(
                f"GROUP BY with more than {cap // 2} groups per workgroup x {n_acc} aggregates exceeds the LDS tier: "
                + self.lib.hs_last_error().decode()
            )
        hs.check(rc, "hs_agg_partial_geom")
        # per-workgroup row ranges + first chunk of every unit, computed by the library, uploaded once
        chunks = np.zeros((max(int(geom.n_chunks), 1), 4), dtype=np.int64)
        chunk0 = np.zeros(n_units + 1, dtype=np.int64)
        hs.check(self.lib.hs_agg_partial_chunks(host_units, n_units, C.byref(geom),
                                                chunks.ctypes.data_as(C.POINTER(hs.hs_chunk)),
                                                chunk0.ctypes.data_as(C.POINTER(C.c_int64))), "hs_agg_partial_chunks")
        if computed:
            # a unit's table holds the groups of ONE unit: sized from the per-unit share of the capacity (x4: open
            # addressing), not from the LDS table that holds every (unit, key) pair - the partial rows' upper bound
            # n_units x unit_cap then stays small enough for the on-chip final merge
            per_unit = max(4, cap // max(batch.n_unit_ids, 1))
            small = 16
            while small < 4 * per_unit:
                small *= 2
            geom.pad = min(int(geom.pad), small)
        unit_cap = int(geom.pad) if shared else cap  # slots per unit of the kernels' output arrays
        if computed:
            n_units = batch.n_unit_ids  # from here on: the unit tables
        slots = n_units * unit_cap
        acc_kinds = [hs.I32 if is_int else hs.F32 for is_int in low.acc_is_int]
        key_idx = low.program.columns[low.key_slot]
        slab = layout = key_out = out_unit = None
        if tail:
            rows = slab_rows if slab_rows is not None else slots
            if slots > rows:
                raise DeviceError(f"exchange slab of {rows} rows cannot hold {slots} partial rows")
            layout, slab, desc = self._tail_slab(batch.cols[key_idx], out_schema, acc_kinds, rows)
            d_unit_ids = self.to_device(np.asarray(batch.unit_ids, dtype=np.int64)) if batch.unit_ids is not None else None
            return {
                "slab": slab, "layout": layout, "desc": desc, "d_unit_ids": d_unit_ids,
                "cols": self._cols_array(batch, low.program.columns, low.program.code_columns), "n_cols": len(low.program.columns),
                "key_slot": low.key_slot, "prog": low.program.to_struct(), "spec": low.spec(), "geom": geom,
                "n_units": n_units, "slots": slots,
                "d_units": self.to_device(chunks.reshape(-1)), "d_chunk0": self.to_device(chunk0),
                "ws": self.workspace(geom.ws_bytes).zero_(),
                "tail": {"desc": desc, "layout": layout, "slab": slab, "agg_to_acc": list(low.agg_to_acc),
                         "acc_kinds": acc_kinds, "key_kind": batch.cols[key_idx].kind,
                         "key_len": batch.cols[key_idx].fixed_len, "n_units": n_units,
                         "key_dict": batch.cols[key_idx].dict},
                "info": {"rows": batch.nrows, "chunks": int(geom.n_chunks), "chunk_rows": int(geom.chunk_rows),
                         "wg_threads": int(geom.wg_threads), "group_cap": cap, "lds_bytes": int(geom.lds_bytes)},
            }
        if slab_rows is not None:
            # multi-GPU: outputs are written straight into the fixed-size slab that gets all-gathered
            from .distributed import SlabLayout  # noqa: PLC0415

            if slots > slab_rows:
                raise DeviceError(f"exchange slab of {slab_rows} rows cannot hold {slots} partial rows")
            kc = batch.cols[key_idx]
            if kc.kind == hs.STR:
                if kc.fixed_len not in (1, 2, 4, 8):
                    raise SlabUnsupported("variable-length string GROUP BY key")
                key_spec = (kc.fixed_len, torch.uint8)
            else:
                key_spec = (hs.KIND_BYTES[kc.kind], _TORCH_DTYPE[kc.kind])
            layout = SlabLayout.build(slab_rows, [key_spec] + [(4, _TORCH_DTYPE[k]) for k in acc_kinds])
            slab = torch.zeros(layout.nbytes, dtype=torch.uint8, device=self.device)
            key_out = layout.column_view(slab, 0)
            acc_bufs = [layout.column_view(slab, 1 + i) for i in range(n_acc)]
            out_unit = layout.order_view(slab)
        else:
            acc_bufs = [self.empty(max(slots, 1), _TORCH_DTYPE[k]) for k in acc_kinds]
        d_unit_ids = None
        if batch.unit_ids is not None:
            d_unit_ids = self.to_device(np.asarray(batch.unit_ids, dtype=np.int64))
            if out_unit is None:
                out_unit = self.empty(max(slots, 1), torch.int64)
        cols_arr, n_cols, unit_slot = (self._cols_array(batch, low.program.columns, low.program.code_columns),
                                       len(low.program.columns), None)
        ws_bytes = geom.ws_bytes
        if computed:
            cols_arr = (hs.hs_col * (n_cols + 1))()
            base_arr = self._cols_array(batch, low.program.columns, low.program.code_columns)
            for slot in range(n_cols):
                cols_arr[slot] = base_arr[slot]
            cols_arr[n_cols] = DCol(hs.U8, batch.unit_col, batch.nrows).as_hs()
            unit_slot, n_cols = n_cols, n_cols + 1
            ws_bytes = ((n_units * unit_cap * 8 + 256 + 15) & ~15) + int(geom.n_chunks) * n_units * unit_cap * max(n_acc, 1) * 8
        return {
            "slab": slab, "layout": layout, "key_out": key_out, "out_unit": out_unit, "d_unit_ids": d_unit_ids,
            "cols": cols_arr, "n_cols": n_cols, "unit_slot": unit_slot,
            "key_slot": low.key_slot, "key_idx": key_idx, "prog": low.program.to_struct(),
            "spec": low.spec(), "geom": geom, "n_units": n_units, "slots": slots, "unit_cap": unit_cap,
            "agg_to_acc": low.agg_to_acc,
            "d_units": self.to_device(chunks.reshape(-1)), "d_chunk0": self.to_device(chunk0),
            "out_rep": self.empty(slots, torch.int64), "out_acc": self.empty(max(slots * n_acc, 1), torch.int64),
            "ngroups": self.empty(max(n_units, 1), torch.int32), "ws": self.workspace(ws_bytes).zero_(),
            "pack_start": self.empty(n_units + 1, torch.int64), "dense_rep": self.empty(max(slots, 1), torch.int64),
            "acc_kinds": acc_kinds, "acc_bufs": acc_bufs,
            "out_ptrs": (C.c_void_p * max(n_acc, 1))(*[t.data_ptr() for t in acc_bufs]),
            "kinds_arr": (C.c_int32 * max(n_acc, 1))(*acc_kinds),
            "info": {"rows": batch.nrows, "chunks": int(geom.n_chunks), "chunk_rows": int(geom.chunk_rows),
                     "wg_threads": int(geom.wg_threads), "group_cap": cap, "lds_bytes": int(geom.lds_bytes),
                     "tier": "shared" if shared else "private"},
        }

    # ---- final merge (A7) ------------------------------------------------------------------------------
    def aggregate_merge(self, batch: DBatch, agg_columns: Sequence[Any], out_schema: Schema,
                        cap_hint: int = 16) -> DBatch:
        """Merge partial rows by key in row (= unit) order; column i+1 is folded with aggregate i's function
        (reference tasks.py:290-292).  Output columns are in-flight (f64 / i64), not rounded; the number
        of groups stays on the device."""
        n_acc = len(agg_columns)
        if n_acc > hs.HS_MAX_ACC:
            raise NotImplementedError(f"more than {hs.HS_MAX_ACC} aggregates")
        spec = hs.hs_agg_spec()
        spec.n_acc = n_acc
        is_int = []
        for i, agg in enumerate(agg_columns):
            spec.op[i] = {"sum": hs.AGG_SUM, "min": hs.AGG_MIN, "max": hs.AGG_MAX}[agg.type]
            integer = batch.cols[i + 1].kind in (hs.I32, hs.I64)
            spec.is_int[i] = 1 if integer else 0
            is_int.append(integer)
        n = batch.nrows
        cap = 4
        while cap < cap_hint:
            cap *= 2
        key = batch.cols[0].as_hs()
        acc_arr = (hs.hs_col * max(n_acc, 1))()
        for i in range(n_acc):
            acc_arr[i] = batch.cols[i + 1].as_hs()
        out_rep = self.empty(cap, torch.int64)
        out_acc = self.empty(max(cap * n_acc, 1), torch.int64)
        ngroups = self.empty(1, torch.int64)
        rc = self.lib.hs_agg_merge(self.stream, C.byref(key), acc_arr, C.byref(spec),
                                   batch.order.data_ptr() if batch.order is not None else None,
                                   batch.total_units or 0, n, batch.n_dev_ptr, cap,
                                   out_rep.data_ptr(), out_acc.data_ptr(), ngroups.data_ptr(), self.flags.data_ptr())
        if rc == 2:
            raise TierExceeded("final merge exceeds the LDS tier: " + self.lib.hs_last_error().decode())
        hs.check(rc, "hs_agg_merge")
        self.last_merge_cap = cap
        n_out = min(cap, n)
        key_col = self.gather_col(batch.cols[0], out_rep, n_out, ngroups)
        if key_col.n != n_out:
            n_out, ngroups_dev = key_col.n, None
        else:
            ngroups_dev = ngroups
        cols = [key_col]
        for i in range(n_acc):
            raw = out_acc[i * cap: i * cap + n_out]
            cols.append(DCol(hs.I64, raw, n_out) if is_int[i] else DCol(hs.F64, raw.view(torch.float64), n_out))
        return DBatch(list(out_schema), cols, n_out, None, ngroups_dev)

    # ---- the short tail: slabs -> result in one launch ------------------------------------------------------
    def aggregate_finish(self, tail: dict, gathered: torch.Tensor, world: int, agg_columns: Sequence[Any],
                         merged_schema: Schema, project: Sequence[Any] | None, out_schema: Schema, cap_hint: int,
                         n_order: int, cache_key: Any = None) -> tuple[list[Any], int, int]:
        """Final merge (reference tasks.py:290-292) + the projection after it (plan.py:190-203: AVG = sum / count,
        renames) + rounding to the stored kinds (io.py:87-94) + the result image, as ONE launch over the partial
        rows of ``world`` slabs; then the query's single device->host copy.  -> (raw columns, nrows, flags)."""
        cap = 4
        while cap < cap_hint:
            cap *= 2
        key = (cache_key, cap, world, gathered.data_ptr(), n_order, tail["slab"].data_ptr(),
               tail["desc"].slab_rows, tail["desc"].stride)
        prep = self._finish_prepared.get(key) if cache_key is not None else None
        if prep is None:
            prep = self._prepare_finish(tail, agg_columns, merged_schema, project, out_schema, cap)
            if cache_key is not None:
                if len(self._finish_prepared) >= 8:
                    self._finish_prepared.pop(next(iter(self._finish_prepared)))
                self._finish_prepared[key] = prep
        p = prep
        desc = p["desc"]
        # the "done" word of a mapped result image is cleared by the host right after it has read the result (finish()
        # below; reset_flags() for runs that never got there): nothing sits between the scan's and this launch
        rc = self.lib.hs_agg_finish(self.stream, gathered.data_ptr(), world, C.byref(desc), C.byref(p["fin"]),
                                    C.byref(p["prog"]) if p["prog"] is not None else None, n_order, cap,
                                    p["result_ptr"], p["scratch"].data_ptr(), self.flags.data_ptr(),
                                    tail["slab"].data_ptr())
        if rc == 2:
            raise TierExceeded("final merge exceeds the LDS tier: " + self.lib.hs_last_error().decode())
        hs.check(rc, "hs_agg_finish")
        self.last_merge_cap = cap
        result, columns, mapped = p["result"], p["columns"], p["host_image"]
        image_bytes = max((off + cap * width for off, _, width in columns), default=16)
        key_dict = tail.get("key_dict")
        stream = torch.cuda.current_stream(self.device)

        key_text = tuple(e.decode("utf-8") for e in key_dict) if key_dict is not None else ()

        def finish() -> tuple[Any, int, int]:
            if mapped is not None:
                # zero-copy hand-over: the launch stored the image into mapped host memory and set "done" last
                done = mapped[4:8].view(np.uint32)
                spins = 0
                while done[0] == 0:
                    spins += 1
                    if spins > 200_000:  # a long scan: let the runtime wait instead of this core
                        stream.synchronize()
                        if done[0] == 0:
                            raise DeviceError("hs_agg_finish completed without handing its result over")
                # a small image (the usual few groups) is copied out whole - the columns below are then views of that
                # one private copy; a large one (thousands of slots) column by column, only the rows that are there
                host = mapped[:image_bytes].copy() if image_bytes <= 8192 else mapped
            else:
                host = result.cpu().numpy()  # the single synchronising copy
            private = host is not mapped
            flags, _, n = _IMAGE_HEADER.unpack_from(host, 0)
            n = min(n, cap)

            def raw_columns() -> list[Any]:
                raw: list[Any] = []
                for off, kind, width in columns:
                    if kind == hs.STR:
                        if key_dict is not None:
                            raw.append(_decode_codes(host[off: off + n], key_dict))
                        else:
                            data = host[off: off + n * width]
                            raw.append(StrCol(np.full(n, width, dtype=np.uint8), data if private else data.copy()))
                    else:
                        data = host[off: off + n * width].view(_NP_DTYPE[kind])
                        raw.append(data if private else data.copy())
                return raw

            if private and 0 < n <= SMALL_RESULT_ROWS:
                # the usual handful of groups: Python values straight from the (private copy of the) image; numpy columns only
                # if the file or the column-wise form is asked for
                py: list[Any] = []
                for off, kind, width in columns:
                    if kind == hs.STR:
                        if key_dict is not None:
                            py.append([key_text[c] for c in host[off: off + n].tolist()])
                        else:
                            text = host[off: off + n * width].tobytes()
                            py.append([text[i * width: (i + 1) * width].decode("utf-8") for i in range(n)])
                    else:
                        vals = _small_struct(kind, n).unpack_from(host, off)
                        py.append([timestamp_to_datetime(v) for v in vals] if kind == hs.I64 else list(vals))
                out: Any = LazyRaw(py, raw_columns)
            else:
                out = raw_columns()
            if mapped is not None:
                done[0] = 0  # ready for the next launch into this image
            return out, n, flags

        if self.rec is not None:
            self.rec.finish = finish
            self.rec.self_cleaning = True  # the launch hands the status words over and zeroes them
            self.rec.keep.append((p, tail, gathered))
        return finish()

    def _prepare_finish(self, tail: dict, agg_columns: Sequence[Any], merged_schema: Schema,
                        project: Sequence[Any] | None, out_schema: Schema, cap: int) -> dict:
        from .lowering import FinishUnsupported, lower_finish  # noqa: PLC0415

        key_kind = tail["key_kind"]
        key_bytes = tail["key_len"] if key_kind == hs.STR else hs.KIND_BYTES[key_kind]
        try:
            fin, prog, outs = lower_finish(tail["agg_to_acc"], tail["acc_kinds"], key_kind, agg_columns, merged_schema,
                                           project, out_schema)
        except FinishUnsupported as e:
            raise TierExceeded(str(e)) from None
        pos = 16
        columns: list[tuple[int, int, int]] = []  # (offset, stored kind, bytes per row)
        for o, (src, index, kind) in enumerate(outs):
            width = key_bytes if src == 0 else hs.KIND_BYTES[kind]
            fin.outs[o].src, fin.outs[o].index, fin.outs[o].kind, fin.outs[o].offset = src, index, kind, pos
            columns.append((pos, kind, width))
            pos = (pos + cap * width + 15) & ~15
        scratch = self.workspace(self.lib.hs_agg_finish_scratch_bytes(cap, fin.n_fold))
        result = host_image = None
        result_ptr = 0
        if self.zero_copy_results:
            # result image in pinned host memory mapped into the device: the launch writes it over PCIe itself
            pinned = torch.zeros(pos + PAD, dtype=torch.uint8).pin_memory()
            dev_ptr = C.c_void_p()
            if self.lib.hs_host_device_pointer(pinned.data_ptr(), C.byref(dev_ptr)) == 0 and dev_ptr.value:
                result, host_image, result_ptr = pinned, pinned.numpy(), dev_ptr.value
        if host_image is None:
            result = torch.zeros(pos + PAD, dtype=torch.uint8, device=self.device)
            result_ptr = result.data_ptr()
        return {"fin": fin, "prog": prog, "desc": tail["desc"], "result": result, "result_ptr": result_ptr,
                "host_image": host_image, "scratch": scratch, "columns": columns}

    # ---- global-memory aggregation tier (any cardinality) ---------------------------------------------------
    def concat_cols(self, parts: Sequence[DCol]) -> DCol:
        """Row-wise concatenation of column pieces (torch.cat = memory plumbing)."""
        if len(parts) == 1:
            return parts[0]
        kind = parts[0].kind
        n = sum(p.n for p in parts)
        if kind == hs.STR and any(p.dict is not None for p in parts):
            if all(p.dict == parts[0].dict for p in parts):  # pieces of one coded column: concatenate the codes
                out = self.empty(n, torch.uint8)
                out.copy_(torch.cat([p.data[: p.n] for p in parts]))
                return DCol(hs.STR, out, n, lens=self.const_lens(1, n), offs=None, fixed_len=1, dict=parts[0].dict)
            parts = [self.decoded(p) for p in parts]
        if kind == hs.STR:
            lens = torch.cat([p.lens[: p.n] for p in parts])
            sizes = [int(p.lens[: p.n].sum().item()) if p.fixed_len < 0 else p.n * p.fixed_len for p in parts]
            data = torch.cat([p.data[:sz] for p, sz in zip(parts, sizes)])
            padded_l, padded_d = self.empty(n, torch.uint8), self.empty(int(sum(sizes)), torch.uint8)
            padded_l.copy_(lens)
            padded_d.copy_(data)
            return self.string_col(padded_l, padded_d, n)
        out = self.empty(n, _TORCH_DTYPE[kind])
        out.copy_(torch.cat([p.data[: p.n] for p in parts]))
        return DCol(kind, out, n)

    def _group_build(self, key: DCol, sel: torch.Tensor | None, n: int):
        cap = 16
        while cap < 2 * max(n, 1):
            cap *= 2
        tkeys = self.empty(cap, torch.int64)
        treps = self.empty(cap, torch.int64)
        slot_start = self.empty(cap + 1, torch.int64)
        positions = self.empty(max(n, 1), torch.int64)
        ws = self.workspace(self.lib.hs_join_build_ws_bytes(n, cap))
        k = key.as_hs()
        hs.check(self.lib.hs_group_build(self.stream, C.byref(k), sel.data_ptr() if sel is not None else None, 0, n, cap,
                                         tkeys.data_ptr(), treps.data_ptr(), slot_start.data_ptr(), positions.data_ptr(),
                                         ws.data_ptr(), self.flags.data_ptr()), "hs_group_build")
        mask = self.empty(cap, torch.uint8)
        hs.check(self.lib.hs_group_mask(self.stream, slot_start.data_ptr(), cap, mask.data_ptr()), "hs_group_mask")
        slot_list = self.empty(cap, torch.int64)
        count = self.empty(1, torch.int64)
        ws2 = self.workspace(self.lib.hs_scan_ws_bytes(cap))
        hs.check(self.lib.hs_compact(self.stream, mask.data_ptr(), cap, slot_list.data_ptr(), count.data_ptr(),
                                     ws2.data_ptr()), "hs_compact")
        return slot_start, positions, slot_list, self.host_int(count)

    def _group_fold(self, vals: Sequence[DCol], ops: Sequence[int], is_int: Sequence[bool], slot_start, positions,
                    slot_list, n_groups: int, sel: torch.Tensor | None, quantise: bool):
        spec = hs.hs_agg_spec()
        spec.n_acc = len(vals)
        for i, (op, integer) in enumerate(zip(ops, is_int)):
            spec.op[i] = op
            spec.is_int[i] = 1 if integer else 0
        arr = (hs.hs_col * max(len(vals), 1))()
        for i, v in enumerate(vals):
            arr[i] = v.as_hs()
        rep_row = self.empty(max(n_groups, 1), torch.int64)
        out_acc = self.empty(max(n_groups * len(vals), 1), torch.int64)
        hs.check(self.lib.hs_group_fold(self.stream, arr, C.byref(spec), slot_list.data_ptr(), n_groups, None,
                                        slot_start.data_ptr(), positions.data_ptr(),
                                        sel.data_ptr() if sel is not None else None, 0, 1 if quantise else 0,
                                        rep_row.data_ptr(), out_acc.data_ptr(), self.flags.data_ptr()), "hs_group_fold")
        cols = []
        for i, integer in enumerate(is_int):
            raw = out_acc[i * n_groups: (i + 1) * n_groups]
            cols.append(DCol(hs.I64, raw, n_groups) if integer else DCol(hs.F64, raw.view(torch.float64), n_groups))
        return rep_row, cols

    def lower_bound(self, sorted_list: torch.Tensor, n: int, queries: torch.Tensor) -> torch.Tensor:
        """out[q] = number of elements of the ascending list sorted_list[:n] that are < queries[q] (device)."""
        out = self.empty(queries.numel(), torch.int64)
        hs.check(self.lib.hs_lower_bound_i64(self.stream, sorted_list.data_ptr() if n > 0 else None, n, None,
                                             queries.data_ptr(), queries.numel(), out.data_ptr()), "hs_lower_bound_i64")
        return out

    # ---- HBM tier by radix partition (csrc/hs_radix.hip) ------------------------------------------------------
    radix_enabled = True  # (tests switch it off to hold the hash-table tier - the form STRING / computed keys keep - to the same rows)

    def _radix_values(self, batch: DBatch, args: Sequence[Any], sel: torch.Tensor | None, n: int):
        """Aggregate arguments for the radix tier -> per argument (value column | None, constant cell, is_int).
        A literal (COUNT's 1) does not travel with the rows; a stored numeric column travels as it is (4 B; gathered through
        the row list behind a WHERE); anything else is evaluated first (8 B cells, one launch for all of them)."""
        import struct  # noqa: PLC0415

        from .lowering import unalias  # noqa: PLC0415

        out: list[Any] = [None] * len(args)
        todo: list[int] = []
        for i, raw in enumerate(args):
            e = unalias(raw)
            cls = type(e).__name__
            if cls == "Lit" and isinstance(e.value, (bool, int)) and -2**63 <= int(e.value) < 2**63:
                out[i] = (None, int(e.value) & (2**64 - 1), True)
            elif cls == "Lit" and isinstance(e.value, float):
                out[i] = (None, struct.unpack("<Q", struct.pack("<d", e.value))[0], False)
            elif cls == "Col" and any(name == e.name for name, _ in batch.schema):
                col = batch.cols[batch.column_index(e.name)]
                if col.kind in (hs.I32, hs.F32, hs.I64) and col.dict is None:
                    # a stored column travels in its stored width (4 B for INTEGER / FLOAT: the tier's fast kernels); behind a
                    # WHERE it is gathered through the row list first - values are indexed by POSITION
                    out[i] = (col if sel is None else self.gather_col(col, sel, n), 0, col.kind != hs.F32)
                else:
                    todo.append(i)
            else:
                todo.append(i)
        if todo:
            for i, (col, tag) in zip(todo, self.eval_numeric(batch, [args[i] for i in todo], sel=sel, n=n)):
                out[i] = (col, 0, tag in ("I", "B"))
        return out

    def sort_by_order(self, order: torch.Tensor, n: int, n_order: int) -> tuple[torch.Tensor, int]:
        """Stable sort of the partial rows by their order key (global block id, -1 = padding) -> (positions in merge
        order with the padding rows dropped, number of rows left)."""
        perm, srt = self.empty(max(n, 1), torch.int64), self.empty(max(n, 1), torch.int64)
        ws = self.workspace(self.lib.hs_sort_by_order_ws_bytes(n))
        hs.check(self.lib.hs_sort_by_order(self.stream, order.data_ptr(), n, max(int(n_order), 1), perm.data_ptr(),
                                           srt.data_ptr(), ws.data_ptr()), "hs_sort_by_order")
        zero = self.to_device(np.zeros(1, dtype=np.int64))
        n_pad = self.host_int(self.lower_bound(srt, n, zero))  # rows with order < 0
        return perm[n_pad:n], n - n_pad

    def unit_ids_per_row(self, unit_rows: Sequence[int], unit_ids: Sequence[int]) -> torch.Tensor:
        """Global block id of every partial row: unit_ids[u] for rows unit_rows[u] .. unit_rows[u + 1]."""
        n = int(unit_rows[-1])
        out = self.empty(max(n, 1), torch.int64)
        bounds = self.to_device(np.asarray(unit_rows, dtype=np.int64))
        ids = self.to_device(np.asarray(unit_ids, dtype=np.int64))
        hs.check(self.lib.hs_expand_by_bounds(self.stream, bounds.data_ptr(), ids.data_ptr(), len(unit_ids), n, out.data_ptr()),
                 "hs_expand_by_bounds")
        return out[:n]

    def group_radix(self, key: DCol, sel: torch.Tensor | None, n: int, bounds: torch.Tensor, n_units: int,
                    max_unit_rows: int, values: Sequence[tuple], ops: Sequence[int], quantise: bool):
        """hs_group_radix_plan / run / emit -> (key column, accumulator columns, groups before every unit) or None when
        a partition outgrew its dictionary (the caller takes the hash-table path)."""
        na = len(values)
        spec = hs.hs_agg_spec()
        spec.n_acc = na
        kinds = (C.c_int32 * max(na, 1))()
        consts = (C.c_uint64 * max(na, 1))()
        cols = (hs.hs_col * max(na, 1))()
        for a, ((col, cell, integer), op) in enumerate(zip(values, ops)):
            spec.op[a] = op
            spec.is_int[a] = 1 if integer else 0
            kinds[a] = col.kind if col is not None else -1
            consts[a] = cell
            cols[a] = col.as_hs() if col is not None else hs.hs_col(hs.U8, -1, None, None, None)
        plan = hs.hs_radix_plan()
        rc = self.lib.hs_group_radix_plan(self.radix_key_code(key), n, n_units, max(int(max_unit_rows), 1), kinds, C.byref(spec),
                                          1 if quantise else 0, C.byref(plan))
        if rc == 2:  # HS_E_LIMIT: not a shape this tier moves (a wide STRING key with MIN / MAX ...): the other HBM tier takes it
            return None
        hs.check(rc, "hs_group_radix_plan")
        ws = self.workspace(self.lib.hs_group_radix_ws_bytes(C.byref(plan)))
        # the group counts per unit and, behind them, the tier's own status word (an overflow here is answered by the other
        # path, not raised): one buffer, so the host reads both with ONE copy
        status = self.empty(n_units + 2, torch.int64)
        unit_groups = status[: n_units + 1]
        radix_flags = status[n_units + 1:].view(torch.int32)[:1]
        status[n_units + 1:].zero_()
        k = key.as_hs()
        hs.check(self.lib.hs_group_radix_run(self.stream, C.byref(plan), C.byref(k), sel.data_ptr() if sel is not None else None,
                                             0, bounds.data_ptr(), cols, consts, C.byref(spec), ws.data_ptr(),
                                             unit_groups.data_ptr(), radix_flags.data_ptr()), "hs_group_radix_run")
        if self.rec is not None:
            self.rec.poisoned = True  # the group count sizes the outputs
        host = status.tolist()  # the one host round trip of the tier
        flags = int(host[-1]) & 0xFFFFFFFF
        if flags & hs.FLAG_DICT_FULL:
            return None
        if flags:
            self.flags[0:1] |= radix_flags  # data-dependent errors surface where the other operators' do
        unit_rows = [int(v) for v in host[: n_units + 1]]
        ng = unit_rows[-1]
        if key.kind == hs.STR:
            out_key = self.empty(max(ng, 1) * key.fixed_len, torch.uint8)
        else:
            out_key = self.empty(max(ng, 1), _TORCH_DTYPE[key.kind])
        okind = [(hs.I32 if integer else hs.F32) if quantise else (hs.I64 if integer else hs.F64) for _, _, integer in values]
        outs = [self.empty(max(ng, 1), _TORCH_DTYPE[kd]) for kd in okind]
        ptrs = (C.c_void_p * max(na, 1))(*[o.data_ptr() for o in outs])
        hs.check(self.lib.hs_group_radix_emit(self.stream, C.byref(plan), ws.data_ptr(), out_key.data_ptr(), ptrs),
                 "hs_group_radix_emit")
        if key.kind == hs.STR:
            lens = torch.full((max(ng, 1),), key.fixed_len, dtype=torch.uint8, device=self.device)
            key_out = DCol(hs.STR, out_key, ng, lens=lens, offs=None, fixed_len=key.fixed_len)
        else:
            key_out = DCol(key.kind, out_key, ng)
        return key_out, [DCol(kd, o, ng) for kd, o in zip(okind, outs)], unit_rows

    @staticmethod
    def radix_key_code(key: DCol) -> int | None:
        """key_kind of hs_group_radix_plan for a GROUP BY column the radix tier takes (its 64-bit key word is the key itself),
        else None: integers, floats, strings of one fixed length <= 16 bytes that are not dictionary codes."""
        if key.dict is not None or key.virtual:
            return None
        if key.kind in (hs.I32, hs.I64, hs.F32, hs.F64):
            return key.kind
        if key.kind == hs.STR and key.fixed_len is not None and 1 <= key.fixed_len <= 16:
            return hs.STR + 256 * key.fixed_len  # <= 7 bytes: one packed key word; 8 .. 16: two words, compared on both
        return None

    def aggregate_partial_global(self, batch: DBatch, filters: Sequence[Any], group_by: Any,
                                 agg_columns: Sequence[Any], out_schema: Schema) -> DBatch:
        """Partial aggregate for any number of groups, ONE pass over all units: an HBM dictionary with one table region
        per unit over the surviving rows' keys, per-group row lists in ascending row order, then one lane per group
        folds the aggregate arguments in row order (bit-identical to the reference's sequential Python sums).  Used
        when the on-chip tiers do not fit.  (Round 1 looped over the units on the host: a dozen launches and two
        host round trips per file block.)"""
        from .lowering import AGG_CODES, expr_key, unalias  # noqa: PLC0415

        batch = self.resolve(batch)
        key_idx = batch.column_index(unalias(group_by).name)
        acc_of: dict[tuple, int] = {}
        args: list[Any] = []
        ops: list[int] = []
        agg_to_acc: list[int] = []
        for agg in agg_columns:
            ident = (agg.type, expr_key(agg.original_col))
            if ident not in acc_of:
                acc_of[ident] = len(args)
                args.append(agg.original_col)
                ops.append(AGG_CODES[agg.type])
            agg_to_acc.append(acc_of[ident])
        n_units = batch.n_units
        d_unit_rows = self.to_device(np.asarray(batch.unit_rows, dtype=np.int64))
        if filters:
            sel, n = self.filter_select(batch, filters)  # ascending row list + its length (the pass's first read-back)
            bounds = self.lower_bound(sel, n, d_unit_rows)  # unit boundaries as POSITIONS in the row list
        else:
            sel, n, bounds = None, batch.nrows, d_unit_rows
        key = batch.cols[key_idx]
        if n == 0:
            empty_key = self.gather_col(key, self.empty(0, torch.int64), 0)
            out_cols = [empty_key] + [DCol(FILE_KIND[t], self.empty(0, _TORCH_DTYPE[FILE_KIND[t]]), 0)
                                      for _, t in out_schema[1:]]
            return DBatch(list(out_schema), out_cols, 0, [0] * (n_units + 1))
        if self.radix_enabled and self.radix_key_code(key) is not None:
            biggest = max(batch.unit_rows[u + 1] - batch.unit_rows[u] for u in range(n_units))
            done = self.group_radix(key, sel, n, bounds, n_units, biggest, self._radix_values(batch, args, sel, n), ops,
                                    quantise=True)
            if done is not None:
                key_col, accs, unit_rows = done
                self.last_global_tier = "radix"
                order = None
                if batch.unit_ids is not None:  # multi-GPU: remember which global unit every partial row came from
                    order = self.unit_ids_per_row(unit_rows, batch.unit_ids)
                return DBatch(list(out_schema), [key_col] + [accs[a] for a in agg_to_acc], unit_rows[-1], unit_rows, order=order,
                              total_units=batch.total_units)
        self.last_global_tier = "hash"
        vals = self.eval_numeric(batch, args, sel=sel, n=n)
        is_int = [tag in ("I", "B") for _, tag in vals]
        # one table region per unit, sized from the unit's UNFILTERED rows (known on the host): >= 2 slots per row
        region_base = [0]
        for u in range(n_units):
            rows_u, size = batch.unit_rows[u + 1] - batch.unit_rows[u], 16
            while size < 2 * rows_u:
                size *= 2
            region_base.append(region_base[-1] + (size if rows_u > 0 else 0))
        cap = max(region_base[-1], 16)
        d_region_base = self.to_device(np.asarray(region_base, dtype=np.int64))
        tkeys, treps = self.empty(cap, torch.int64), self.empty(cap, torch.int64)
        slot_start, positions = self.empty(cap + 1, torch.int64), self.empty(max(n, 1), torch.int64)
        ws = self.workspace(self.lib.hs_join_build_ws_bytes(n, cap))
        k = key.as_hs()
        hs.check(self.lib.hs_group_build_units(self.stream, C.byref(k), sel.data_ptr() if sel is not None else None, 0, n,
                                               bounds.data_ptr(), d_region_base.data_ptr(), n_units, cap, tkeys.data_ptr(),
                                               treps.data_ptr(), slot_start.data_ptr(), positions.data_ptr(), ws.data_ptr(),
                                               self.flags.data_ptr()), "hs_group_build_units")
        mask = self.empty(cap, torch.uint8)
        hs.check(self.lib.hs_group_mask(self.stream, slot_start.data_ptr(), cap, mask.data_ptr()), "hs_group_mask")
        slot_list, d_groups = self.empty(cap, torch.int64), self.empty(1, torch.int64)
        ws2 = self.workspace(self.lib.hs_scan_ws_bytes(cap))
        hs.check(self.lib.hs_compact(self.stream, mask.data_ptr(), cap, slot_list.data_ptr(), d_groups.data_ptr(),
                                     ws2.data_ptr()), "hs_compact")
        ng = self.host_int(d_groups)  # sizes the outputs (second read-back)
        rep_row, cols = self._group_fold([v for v, _ in vals], ops, is_int, slot_start, positions, slot_list, ng, sel,
                                         quantise=True)
        key_col = self.gather_col(key, rep_row, ng)
        types = [ColumnType.INTEGER if i else ColumnType.FLOAT for i in is_int]
        accs = self.quantise_cols(cols, types)
        # groups per unit: the slot list is ascending, a unit's groups sit in its region
        unit_rows = [int(v) for v in self.lower_bound(slot_list, ng, d_region_base).tolist()]
        order = None
        if batch.unit_ids is not None:  # multi-GPU: remember which global unit every partial row came from
            order = self.unit_ids_per_row(unit_rows, batch.unit_ids)
        return DBatch(list(out_schema), [key_col] + [accs[a] for a in agg_to_acc], ng, unit_rows, order=order,
                      total_units=batch.total_units)

    def aggregate_merge_global(self, batch: DBatch, agg_columns: Sequence[Any], out_schema: Schema) -> DBatch:
        """Final merge for any number of groups: dictionary over the partial rows' keys, row lists in merge
        order, one lane per group folds its partials front to back (the reference's order)."""
        batch = self.resolve(batch)
        n = batch.nrows
        sel = None
        if batch.order is not None:  # multi-GPU: visit rows by (block id, row); padding rows (order < 0) dropped
            n_order = batch.total_units if batch.total_units else 1 << 31  # unknown: sort on all 32 bits
            sel, n = self.sort_by_order(batch.order, n, n_order)
        ops = [{"sum": hs.AGG_SUM, "min": hs.AGG_MIN, "max": hs.AGG_MAX}[a.type] for a in agg_columns]
        vals = batch.cols[1: 1 + len(agg_columns)]
        is_int = [v.kind in (hs.I32, hs.I64) for v in vals]
        key = batch.cols[0]
        if sel is not None:  # values are read by POSITION: bring them into visiting order
            vals = [self.gather_col(v, sel, n) for v in vals]
        if (self.radix_enabled and n > 0 and self.radix_key_code(key) is not None
                and all(v.kind in (hs.I32, hs.F32, hs.I64, hs.F64) and v.dict is None for v in vals)):
            # the partial rows in merge order (single GPU: as they are; multi-rank: through sel) are ONE unit of n rows
            bounds = self.to_device(np.asarray([0, n], dtype=np.int64))
            done = self.group_radix(key, sel, n, bounds, 1, n, [(v, 0, i) for v, i in zip(vals, is_int)], ops, quantise=False)
            if done is not None:
                key_col, cols, unit_rows = done
                return DBatch(list(out_schema), [key_col] + cols, unit_rows[-1], [0, unit_rows[-1]])
        slot_start, positions, slot_list, ng = self._group_build(batch.cols[0], sel, n)
        rep_row, cols = self._group_fold(vals, ops, is_int, slot_start, positions, slot_list, ng, sel, quantise=False)
        key_col = self.gather_col(batch.cols[0], rep_row, ng)
        return DBatch(list(out_schema), [key_col] + cols, ng, [0, ng])

    # ---- hash partitioning (A6/A9) -------------------------------------------------------------------------
    def partition(self, batch: DBatch, key_index: int, n_parts: int) -> tuple[torch.Tensor, list[int]]:
        """Stable partition of the batch's rows by ``hash(key) % n_parts`` -> (perm, part_start)."""
        n = batch.nrows
        part = self.empty(n, torch.uint8)
        key = batch.cols[key_index].as_hs()
        hs.check(self.lib.hs_partition_ids(self.stream, C.byref(key), None, n, n_parts, part.data_ptr()),
                 "hs_partition_ids")
        perm = self.empty(n, torch.int64)
        part_start = self.empty(n_parts + 1, torch.int64)
        ws = self.workspace(self.lib.hs_partition_ws_bytes(n, n_parts))
        hs.check(self.lib.hs_partition_perm(self.stream, part.data_ptr(), n, n_parts, perm.data_ptr(),
                                            part_start.data_ptr(), ws.data_ptr()), "hs_partition_perm")
        if self.rec is not None:
            self.rec.poisoned = True  # partition sizes reach the host
        return perm, [int(v) for v in part_start.tolist()]

    def partition_by_ids(self, ids: torch.Tensor, n: int, n_parts: int) -> tuple[torch.Tensor, list[int]]:
        """Stable counting sort of rows by a u8 id per row -> (perm, start)."""
        perm = self.empty(n, torch.int64)
        part_start = self.empty(n_parts + 1, torch.int64)
        ws = self.workspace(self.lib.hs_partition_ws_bytes(n, n_parts))
        hs.check(self.lib.hs_partition_perm(self.stream, ids.data_ptr(), n, n_parts, perm.data_ptr(),
                                            part_start.data_ptr(), ws.data_ptr()), "hs_partition_perm")
        if self.rec is not None:
            self.rec.poisoned = True
        return perm, [int(v) for v in part_start.tolist()]

    # ---- device-side packing of the row exchange between ranks ---------------------------------------------------
    torch = torch  # (the engine's exchange code names dtypes without importing torch itself)

    @staticmethod
    def torch_dtype(kind: int) -> torch.dtype:
        return _TORCH_DTYPE[kind]

    def partition_by_ids_dev(self, ids: torch.Tensor, n: int, n_parts: int) -> tuple[torch.Tensor, torch.Tensor]:
        """partition_by_ids without the read-back: -> (perm, start int64[n_parts + 1] on the device)."""
        perm = self.empty(max(n, 1), torch.int64)
        part_start = self.empty(n_parts + 1, torch.int64)
        ws = self.workspace(self.lib.hs_partition_ws_bytes(n, n_parts))
        hs.check(self.lib.hs_partition_perm(self.stream, ids.data_ptr() if n > 0 else None, n, n_parts, perm.data_ptr(),
                                            part_start.data_ptr(), ws.data_ptr()), "hs_partition_perm")
        return perm, part_start

    def permute_col(self, col: DCol, perm: torch.Tensor, n: int) -> DCol:
        """col[perm[i]] for a permutation of ALL n rows.  Unlike gather_col the host learns nothing: the payload of a
        variable-length STRING column is exactly as long as the source's, so its buffer is sized from that."""
        if col.kind != hs.STR or col.fixed_len in (1, 2, 4, 8):
            return self.gather_col(col, perm, n)
        src = col.as_hs()
        lens = self.empty(n, torch.uint8)
        hs.check(self.lib.hs_gather_str_lens(self.stream, C.byref(src), col.n, perm.data_ptr(), n, lens.data_ptr(),
                                             self.flags.data_ptr()), "hs_gather_str_lens")
        offs = self.empty(n + 1, torch.int64)
        minmax = self.empty(2, torch.int32)
        ws = self.workspace(self.lib.hs_scan_ws_bytes(n))
        hs.check(self.lib.hs_str_offsets(self.stream, lens.data_ptr(), n, offs.data_ptr(), minmax.data_ptr(), ws.data_ptr()),
                 "hs_str_offsets")
        data = self.empty(int(col.data.numel()), torch.uint8)
        hs.check(self.lib.hs_gather_str_bytes(self.stream, C.byref(src), col.n, perm.data_ptr(), n, offs.data_ptr(),
                                              data.data_ptr()), "hs_gather_str_bytes")
        if col.fixed_len >= 0:  # every row has that length: the payload is n x fixed_len bytes, no offsets needed
            return DCol(hs.STR, data, n, lens=lens, offs=None, fixed_len=col.fixed_len)
        return DCol(hs.STR, data, n, lens=lens, offs=offs, fixed_len=-1)

    def exchange_boundaries(self, start_dev: torch.Tensor, world: int, offs_list: Sequence[torch.Tensor]):
        """The sizes the host needs to lay out an exchange, in ONE device->host copy: the destination boundaries
        start[0 .. world] and, per variable-length payload, its byte offsets at those boundaries."""
        k = len(offs_list)
        out = self.empty((world + 1) * (1 + k), torch.int64)
        out[: world + 1].copy_(start_dev[: world + 1])
        for j, offs in enumerate(offs_list):
            dst = out[(world + 1) * (1 + j): (world + 1) * (2 + j)]
            hs.check(self.lib.hs_gather_fixed(self.stream, offs.data_ptr(), 8, offs.numel(), start_dev.data_ptr(), world + 1,
                                              None, dst.data_ptr(), self.flags.data_ptr()), "hs_gather_fixed")
        if self.rec is not None:
            self.rec.poisoned = True  # sizes reach the host
        flat = out.tolist()
        start = [int(v) for v in flat[: world + 1]]
        var = [[int(v) for v in flat[(world + 1) * (1 + j): (world + 1) * (2 + j)]] for j in range(k)]
        return start, var

    def copy_segments(self, segs: Sequence[tuple[int, int, int]]) -> None:
        """dst <- src for a list of (source address, destination address, bytes): one launch (hs_copy_segments)."""
        if not segs:
            return
        arr = np.zeros((len(segs), 3), dtype=np.int64)
        arr[:] = segs
        recording, self.rec = self.rec, None  # (an exchange is never part of a recording: its sizes reached the host)
        try:
            d_segs = self.to_device(arr.reshape(-1))
        finally:
            self.rec = recording
        hs.check(self.lib.hs_copy_segments(self.stream, d_segs.data_ptr(), len(segs), int(arr[:, 2].max())), "hs_copy_segments")
        if self.rec is not None:
            self.rec.keep.append(d_segs)

    def partition_ids(self, batch: DBatch, key_index: int, n_parts: int) -> torch.Tensor:
        part = self.empty(batch.nrows, torch.uint8)
        key = batch.cols[key_index].as_hs()
        hs.check(self.lib.hs_partition_ids(self.stream, C.byref(key), None, batch.nrows, n_parts, part.data_ptr()),
                 "hs_partition_ids")
        return part

    # ---- primary-key / foreign-key join: rows stay in place, units are computed -----------------
    DIRECT_JOIN_MAX_SPREAD = 16  # direct addressing while the build side's key range is <= this many slots per key

    def join_probe_unique(self, build_key: DCol, probe_key: DCol, n_parts: int, payload: DCol | None = None,
                          want_rows: bool = True):
        """Inner equi-join of INTEGER keys whose build side is expected to hold every key once (HS_FLAG_JOIN_DUP in
        the status word otherwise: the engine then re-runs the query through the general join).  ->
        (build_row int64[n_probe] or None, unit u8[n_probe] with 0xff = no match, payload u8[n_probe] or None)."""
        n_build, n_probe = build_key.n, probe_key.n
        # the key range of a column is a property of the column: learnt once (one small read-back), so the join itself
        # runs without the host learning anything - and can be recorded and replayed like any other query
        span = build_key.__dict__.get("_hs_minmax")
        if span is None:
            minmax = self.empty(2, torch.int32)
            hs.check(self._raw_lib.hs_minmax_i32(self.stream, build_key.data.data_ptr(), n_build, minmax.data_ptr()), "hs_minmax_i32")
            if self.rec is not None:
                self.rec.poisoned = True  # this run learnt the range on the way; the next one finds it cached
            span = build_key.__dict__["_hs_minmax"] = tuple(int(v) for v in minmax.tolist())
        lo, hi = span
        spread = hi - lo + 1 if n_build else 1
        direct = n_build > 0 and spread <= self.DIRECT_JOIN_MAX_SPREAD * n_build
        if direct:
            slots, key_min = spread, lo
        else:
            slots, key_min = 16, 0
            while slots < 2 * max(n_build, 1):
                slots *= 2
        table = self.empty(slots + 8, torch.int32)  # + the occupied-slot counter of the direct build
        if self.join_events is not None:
            self.op(self.join_events[0].record)
        hs.check(self.lib.hs_join_build_unique(self.stream, build_key.data.data_ptr(), n_build, key_min, slots,
                                               1 if direct else 0, table.data_ptr(), self.flags.data_ptr()),
                 "hs_join_build_unique")
        rows = self.empty(n_probe, torch.int64) if want_rows else None
        unit = self.empty(n_probe, torch.uint8)
        pay = self.empty(n_probe, torch.uint8) if payload is not None else None
        hs.check(self.lib.hs_join_probe_unique(self.stream, probe_key.data.data_ptr(), n_probe, build_key.data.data_ptr(),
                                               key_min, slots, 1 if direct else 0, table.data_ptr(), n_parts,
                                               rows.data_ptr() if rows is not None else None, unit.data_ptr(),
                                               payload.data.data_ptr() if payload is not None else None,
                                               pay.data_ptr() if pay is not None else None), "hs_join_probe_unique")
        if self.join_events is not None:
            self.op(self.join_events[1].record)
        self.last_join = {"mode": "direct" if direct else "hashed", "slots": slots, "n_build": n_build, "n_probe": n_probe}
        return rows, unit, pay

    # ---- the join's probe inside the aggregate scan (round 3) -----------------------------------------
    JOIN8_MAX_SPREAD = 32      # the byte table is used while the build side's key range holds <= this many slots per key
    JOIN8_MAX_SLOTS = 1 << 30

    def key_range(self, col: DCol) -> tuple[int, int, int]:
        """(min, max, rows) of an INTEGER column, learnt once per column object (one small read-back)."""
        span = col.__dict__.get("_hs_minmax")
        if span is None:
            if col.n == 0:
                span = (2**31 - 1, -(2**31))
            else:
                minmax = self.empty(2, torch.int32)
                hs.check(self._raw_lib.hs_minmax_i32(self.stream, col.data.data_ptr(), col.n, minmax.data_ptr()), "hs_minmax_i32")
                if self.rec is not None:
                    self.rec.poisoned = True  # this run learnt the range on the way; the next one finds it cached
                span = tuple(int(v) for v in minmax.tolist())
            col.__dict__["_hs_minmax"] = span
        return span[0], span[1], col.n

    def join8_plan(self, build_key: DCol, dist_ctx: tuple | None = None) -> dict | None:
        """Shape of the byte table for this build key column - or None when the key range is too sparse / wide for direct
        addressing.  N ranks: every rank's (min, max, rows) travel once per column (cached), so that all ranks take the
        same decision and know the segment length of the gathered build side."""
        plan = build_key.__dict__.get("_hs_join8_plan")
        if plan is not None and plan["world"] == (dist_ctx[2] if dist_ctx else 1):
            return plan["shape"]
        lo, hi, n = self.key_range(build_key)
        rows = [n]
        if dist_ctx is not None:
            dist, group, world = dist_ctx
            mine = torch.tensor([lo, hi, n], dtype=torch.int64)
            everyone = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
            if dist.get_backend(group) == "gloo":
                dist.all_gather(everyone, mine, group=group)
            else:
                dev_all = [t.to(self.device) for t in everyone]
                dist.all_gather(dev_all, mine.to(self.device), group=group)
                everyone = [t.cpu() for t in dev_all]
            if self.rec is not None:
                self.rec.poisoned = True
            triples = [[int(v) for v in t.tolist()] for t in everyone]
            lo, hi = min(t[0] for t in triples), max(t[1] for t in triples)
            rows = [t[2] for t in triples]
        total = sum(rows)
        slots = hi - lo + 1 if total else 1
        shape = None
        if 0 < total < 0xFFFFFFFF and 0 < slots <= min(self.JOIN8_MAX_SLOTS, self.JOIN8_MAX_SPREAD * total) and lo >= -(2**31):
            seg_len = (max(rows) + 15) & ~15
            shape = {"key_min": lo, "slots": slots, "rows": rows, "seg_len": seg_len, "total": total}
        build_key.__dict__["_hs_join8_plan"] = {"world": dist_ctx[2] if dist_ctx else 1, "shape": shape}
        return shape

    def join8_table(self, shape: dict, build_key: DCol, payload: DCol | None, n_parts: int,
                    dist_ctx: tuple | None = None) -> dict:
        """Build the join's byte table (hs_join8_build: window partition + assembly in LDS, the table is written once,
        coalesced).  N ranks: the build side is small next to the probe side (orders : lineitem = 1 : 4 rows, 5 : 12
        referenced bytes) - every rank all-gathers key + payload code columns (padded to one segment length, valid
        counts alongside) and builds the WHOLE table; its probe rows then never leave the rank.  Duplicate keys raise
        HS_FLAG_JOIN_DUP in the status word."""
        slots, key_min = shape["slots"], shape["key_min"]
        table = self.workspace(self.lib.hs_join8_table_bytes(slots))
        keys, codes, n, seg_len, counts = build_key.data, payload.data if payload is not None else None, build_key.n, 0, None
        if dist_ctx is not None:
            from .distributed import all_gather_into  # noqa: PLC0415

            dist, group, world = dist_ctx
            seg_len = shape["seg_len"]
            timed = self.exchange_events
            if timed is not None:
                self.op(timed[0].record)
            kpad = self.empty(seg_len, torch.int32)
            self.op(kpad[:n].copy_, keys[:n])
            keys = self.empty(world * seg_len, torch.int32)
            self.op(all_gather_into, dist, kpad, keys, group)
            if codes is not None:
                cpad = self.empty(seg_len, torch.uint8)
                self.op(cpad[:n].copy_, codes[:n])
                codes = self.empty(world * seg_len, torch.uint8)
                self.op(all_gather_into, dist, cpad, codes, group)
            if timed is not None:
                self.op(timed[1].record)
            # valid rows per segment: known from the plan's exchange (a table column's row count does not change)
            counts = self.to_device_const(np.asarray(shape["rows"], dtype=np.int64))
            n = world * seg_len
        ws = self.workspace(self.lib.hs_join8_ws_bytes(n, slots))
        if self.join_events is not None:  # the build kernels alone (the gather above is timed as exchange)
            self.op(self.join_events[0].record)
        hs.check(self.lib.hs_join8_build(self.stream, keys.data_ptr(), codes.data_ptr() if codes is not None else None, n,
                                         seg_len, counts.data_ptr() if counts is not None else None, key_min, slots,
                                         table.data_ptr(), ws.data_ptr(), self.flags.data_ptr()), "hs_join8_build")
        if self.join_events is not None:
            self.op(self.join_events[1].record)
        self.last_join = {"mode": "byte table", "slots": slots, "n_build": shape["total"], "table_bytes": int(table.numel())}
        return {"table": table, "key_min": key_min, "slots": slots, "n_parts": n_parts, "keep": (keys, codes, counts, ws)}

    # ---- N ranks: the byte table sharded by the probe side's key stripes (round 4) -----------------------
    JOIN8_MAX_STRIPES = 4096

    def join8_stripes(self, probe_key: DCol, unit_rows: Sequence[int], unit_ids: Sequence[int] | None, shape: dict,
                      dist_ctx: tuple) -> dict | None:
        """The key stripe [min, max] of every probe-side block on every rank - or None when the probe table is not clustered
        on the key (stripes of consecutive blocks must not step backwards), in which case the caller keeps the all-gathered
        build.  Learnt once per probe column (one launch + one small read-back + one object all-gather: a collective moment,
        like the plan exchange of join8_plan; the answer is the same on every rank because it is computed from the gathered
        list)."""
        dist, group, world = dist_ctx
        cached = probe_key.__dict__.get("_hs_join8_stripes")
        stamp = (world, tuple(unit_rows), tuple(unit_ids or ()), shape["key_min"], shape["slots"])
        if cached is not None and cached["stamp"] == stamp:
            return cached["stripes"]
        n_units = len(unit_rows) - 1
        ids = list(unit_ids) if unit_ids is not None else list(range(n_units))
        mine: list[tuple[int, int, int]] = []
        if n_units > 0 and probe_key.n > 0:
            bounds = self.to_device_const(np.asarray(unit_rows, dtype=np.int64))
            minmax = self.empty(2 * n_units, torch.int32)
            hs.check(self._raw_lib.hs_minmax_i32_units(self.stream, probe_key.data.data_ptr(), bounds.data_ptr(), n_units,
                                                       minmax.data_ptr()), "hs_minmax_i32_units")
            vals = minmax.tolist()
            mine = [(int(ids[u]), int(vals[2 * u]), int(vals[2 * u + 1])) for u in range(n_units)
                    if unit_rows[u + 1] > unit_rows[u]]
        if self.rec is not None:
            self.rec.poisoned = True  # this run learnt the stripes on the way; the next one finds them cached
        everyone: list = [None] * world
        dist.all_gather_object(everyone, mine, group=group)
        blocks = sorted((b, lo, hi, r) for r, part in enumerate(everyone) for b, lo, hi in part)
        stripes = None
        clustered = all(blocks[i][1] <= blocks[i + 1][1] and blocks[i][2] <= blocks[i + 1][2] for i in range(len(blocks) - 1))
        if clustered and 0 < len(blocks) <= self.JOIN8_MAX_STRIPES and world <= 64:
            key_min, slots = shape["key_min"], shape["slots"]
            n_win = (slots + hs.JOIN8_WINDOW - 1) // hs.JOIN8_WINDOW
            mask = np.zeros(n_win, dtype=np.uint8)
            rank = dist.get_rank(group)
            for _b, lo, hi, r in blocks:
                if r == rank:
                    w0, w1 = max(lo - key_min, 0) // hs.JOIN8_WINDOW, min(hi - key_min, slots - 1) // hs.JOIN8_WINDOW
                    if w1 >= w0:
                        mask[w0: w1 + 1] = 1
            arr = np.asarray(blocks, dtype=np.int64)
            stripes = {"n": len(blocks), "min": self.to_device_const(arr[:, 1].astype(np.int32)),
                       "max": self.to_device_const(arr[:, 2].astype(np.int32)),
                       "owner": self.to_device_const(arr[:, 3].astype(np.int32)),
                       "window_mask": self.to_device_const(mask), "windows": int(mask.sum()), "n_windows": int(n_win)}
        probe_key.__dict__["_hs_join8_stripes"] = {"stamp": stamp, "stripes": stripes}
        return stripes

    def join8_table_sharded(self, shape: dict, stripes: dict, build_key: DCol, payload: DCol | None, n_parts: int,
                            dist_ctx: tuple) -> dict:
        """The byte table on N ranks WITHOUT replicating the build (reference: both inputs shuffled on the key,
        plan.py:186-189; one JoinJob per partition, plan.py:99-109).  Every rank routes its build rows (key + payload code)
        to the rank(s) whose probe blocks' key stripes contain the key - two all_to_all_single calls of ~1/world of the
        build side each way instead of all-gathering all of it -, builds only the windows of the table its own stripes
        reach and probes in place.  The split sizes are data: the first run reads them back and agrees the receive sizes
        (one small collective); later runs - recorded and replayed - pass the same sizes while the routing kernel verifies
        on the device that the data still routes that way (HS_FLAG_ROUTE_STALE -> the engine forgets the sizes and repeats
        the query)."""
        from .distributed import _a2a, exchange_size_matrix  # noqa: PLC0415

        dist, group, world = dist_ctx
        slots, key_min, n = shape["slots"], shape["key_min"], build_key.n
        table = self.workspace(self.lib.hs_join8_table_bytes(slots))
        route_ws = self.workspace(self.lib.hs_join8_route_ws_bytes(n, world))
        timed = self.exchange_events
        if timed is not None:
            self.op(timed[0].record)
        sizes = build_key.__dict__.get("_hs_join8_route")
        stamp = (world, id(stripes), payload is not None)
        dest_start = self.empty(world + 1, torch.int32)
        hs.check(self.lib.hs_join8_route_count(self.stream, build_key.data.data_ptr(), n, stripes["min"].data_ptr(),
                                               stripes["max"].data_ptr(), stripes["owner"].data_ptr(), stripes["n"], world,
                                               route_ws.data_ptr(), dest_start.data_ptr()), "hs_join8_route_count")
        if sizes is None or sizes["stamp"] != stamp:
            starts = [int(v) & 0xFFFFFFFF for v in dest_start.tolist()]
            if self.rec is not None:
                self.rec.poisoned = True  # sizes learnt on the host + a collective of its own: never part of a replay
            send = [starts[d + 1] - starts[d] for d in range(world)]
            recv = [row[0] for row in exchange_size_matrix(dist, [[c] for c in send], self.device, group)]
            expect = torch.tensor(starts, dtype=torch.int64).to(torch.int32).to(self.device)
            sizes = {"stamp": stamp, "send": send, "recv": recv, "expect": expect, "stripes": stripes}
            build_key.__dict__["_hs_join8_route"] = sizes
            self.__dict__.setdefault("_route_caches", []).append(build_key)
        send, recv = sizes["send"], sizes["recv"]
        n_out, n_in = sum(send), sum(recv)
        out_keys = self.empty(max(n_out, 1), torch.int32)  # (torch hands out a null pointer for an empty tensor)
        out_codes = self.empty(max(n_out, 1), torch.uint8) if payload is not None else None
        hs.check(self.lib.hs_join8_route(self.stream, build_key.data.data_ptr(), payload.data.data_ptr() if payload is not None else None,
                                         n, stripes["min"].data_ptr(), stripes["max"].data_ptr(), stripes["owner"].data_ptr(),
                                         stripes["n"], world, route_ws.data_ptr(), dest_start.data_ptr(), sizes["expect"].data_ptr(),
                                         out_keys.data_ptr(),
                                         out_codes.data_ptr() if out_codes is not None else None, n_out, self.flags.data_ptr()),
                 "hs_join8_route")
        keys = self.empty(max(n_in, 1), torch.int32)
        self.op(_a2a, dist, keys[:n_in], out_keys[:n_out], recv, send, group)
        codes = None
        if out_codes is not None:
            codes = self.empty(max(n_in, 1), torch.uint8)
            self.op(_a2a, dist, codes[:n_in], out_codes[:n_out], recv, send, group)
        if timed is not None:
            self.op(timed[1].record)
        ws = self.workspace(self.lib.hs_join8_ws_bytes(n_in, slots))
        if self.join_events is not None:
            self.op(self.join_events[0].record)
        hs.check(self.lib.hs_join8_build_windows(self.stream, keys.data_ptr(), codes.data_ptr() if codes is not None else None, n_in,
                                                 key_min, slots, stripes["window_mask"].data_ptr(), table.data_ptr(), ws.data_ptr(),
                                                 self.flags.data_ptr()), "hs_join8_build_windows")
        if self.join_events is not None:
            self.op(self.join_events[1].record)
        row_bytes = 4 + (1 if payload is not None else 0)
        self.last_join = {"mode": "byte table", "sharded": True, "slots": slots, "n_build": shape["total"],
                          "build_rows_local": n, "build_rows_received": n_in, "bytes_sent": (n_out - send[dist.get_rank(group)]) * row_bytes,
                          "bytes_received": (n_in - recv[dist.get_rank(group)]) * row_bytes,
                          "table_bytes": stripes["windows"] * hs.JOIN8_WINDOW, "table_address_range": int(table.numel()),
                          "windows": f"{stripes['windows']} of {stripes['n_windows']}"}
        return {"table": table, "key_min": key_min, "slots": slots, "n_parts": n_parts,
                "keep": (keys, codes, out_keys, out_codes, route_ws, dest_start, ws, sizes)}

    def forget_routes(self) -> None:
        """HS_FLAG_ROUTE_STALE: the cached split sizes of every sharded build are dropped (the next run agrees them anew)."""
        for col in self.__dict__.pop("_route_caches", []):
            col.__dict__.pop("_hs_join8_route", None)

    def to_device_const(self, arr: np.ndarray) -> torch.Tensor:
        """A small constant of the query SHAPE (not of this run's data): uploading it does not poison a recording."""
        recording, self.rec = self.rec, None
        try:
            t = self.to_device(arr)
        finally:
            self.rec = recording
        if self.rec is not None:
            self.rec.keep.append(t)
        return t

    def aggregate_join8(self, batch: DBatch, filters: Sequence[Any], group_by: Any, agg_columns: Sequence[Any],
                        out_schema: Schema, group_cap_hint: int, cache_key: Any = None,
                        dist_ctx: tuple | None = None, raw_tables: list | None = None) -> Any:
        """Partial aggregate per JoinJob (plan.py:99-109) of a join whose rows were never produced: the scan looks every
        probe key up in the join's byte table (`batch.join8`), takes the unit from python_hash(key) % partitions and
        the GROUP BY key / nothing from the table byte, and folds the probe side's columns into per-(unit, key) cells
        (shared-dictionary tier with computed units).  The units' RAW tables - on N ranks added up over the ranks first
        (one small all-gather), because the reference rounds a JoinJob's sums once, over all of its rows - are rounded
        into an exchange slab, which the short tail's finish launch turns into the result.  Raises TierExceeded for
        shapes this path does not hold."""
        j = batch.join8
        low = lower_aggregate(batch.schema, batch.kinds, filters, group_by, agg_columns, batch.dicts)
        n_cols = len(low.program.columns)
        if low.numeric_slots >= hs.HS_FUSED_COLS or n_cols >= hs.HS_FUSED_COLS:
            raise TierExceeded("no preloaded column slot left for the unit ids")
        if low.program.code_columns:
            raise TierExceeded("predicates on dictionary codes next to the fused probe")
        key_idx = low.program.columns[low.key_slot]
        kc = batch.cols[key_idx]
        key_ok = kc.virtual == hs.JOIN8_CODE or (kc.virtual == 0 and (kc.kind == hs.I32 or (kc.kind == hs.STR and kc.fixed_len in (1, 2, 4))))
        if not key_ok or any(batch.cols[c].virtual and c != key_idx for c in low.program.columns):
            raise TierExceeded("the fused probe needs a GROUP BY key of at most 4 bytes and no other use of the build side")
        n_acc = len(low.acc_ops)
        n_units = int(j["n_parts"])
        cap = 16
        while cap < max(int(group_cap_hint), 4) * n_units:
            cap *= 2
        cap = min(cap, 4096)
        world = dist_ctx[2] if dist_ctx else 1
        key = (cache_key, cap, batch.nrows, j["table"].data_ptr(), j["key_min"], j["slots"], world,
               tuple((c.kind, c.virtual, c.fixed_len, c.data.data_ptr(), c.n) for c in batch.cols))
        preps = self.__dict__.setdefault("_join8_prepared", {})
        p = preps.get(key) if cache_key is not None else None
        if p is None:
            host_units = (C.c_int64 * 2)(0, batch.nrows)
            geom = hs.hs_agg_geom()
            rc = self.lib.hs_agg_shared_geom(host_units, 1, n_acc, cap, C.byref(geom))
            if rc == 2:
                raise TierExceeded("fused probe: " + self.lib.hs_last_error().decode())
            hs.check(rc, "hs_agg_shared_geom")
            chunks = np.zeros((max(int(geom.n_chunks), 1), 4), dtype=np.int64)
            chunk0 = np.zeros(2, dtype=np.int64)
            hs.check(self.lib.hs_agg_partial_chunks(host_units, 1, C.byref(geom), chunks.ctypes.data_as(C.POINTER(hs.hs_chunk)),
                                                    chunk0.ctypes.data_as(C.POINTER(C.c_int64))), "hs_agg_partial_chunks")
            per_unit, small = max(4, cap // n_units), 16
            while small < 4 * per_unit:
                small *= 2
            geom.pad = min(int(geom.pad), small)  # slots of ONE unit's table (x4: open addressing)
            unit_cap = int(geom.pad)
            slots = n_units * unit_cap
            acc_kinds = [hs.I32 if is_int else hs.F32 for is_int in low.acc_is_int]
            key_stored = DCol(hs.STR, kc.data, kc.n, fixed_len=1, dict=kc.dict) if kc.virtual else kc
            layout, slab, desc = self._tail_slab(key_stored, out_schema, acc_kinds, slots)
            cols_arr = (hs.hs_col * (n_cols + 1))()
            base_arr = self._cols_array(batch, low.program.columns)
            for slot in range(n_cols):
                cols_arr[slot] = base_arr[slot]
            cols_arr[n_cols] = DCol(hs.U8, j["probe_key"].data, batch.nrows, virtual=hs.JOIN8_UNIT).as_hs()
            table_bytes = 16 + slots * 8 * (1 + n_acc)
            xbuf = torch.zeros(table_bytes + PAD, dtype=torch.uint8, device=self.device)
            join = hs.hs_join8(j["table"].data_ptr(), j["slots"], j["key_min"], n_units)
            p = {"geom": geom, "d_units": self.to_device_const(chunks.reshape(-1)), "cols": cols_arr, "n_cols": n_cols + 1,
                 "unit_slot": n_cols, "key_slot": low.key_slot, "prog": low.program.to_struct(), "spec": low.spec(),
                 "join": join, "n_units": n_units, "unit_cap": unit_cap, "slots": slots, "xbuf": xbuf,
                 "table_bytes": table_bytes, "out_rep": self.empty(slots, torch.int64),
                 "ws": self.workspace(max(int(geom.n_chunks), 1) * slots * max(n_acc, 1) * 8),
                 "slab": slab, "layout": layout, "desc": desc, "keep": (batch, j),
                 "tail": {"desc": desc, "layout": layout, "slab": slab, "agg_to_acc": list(low.agg_to_acc),
                          "acc_kinds": acc_kinds, "key_kind": key_stored.kind, "key_len": key_stored.fixed_len,
                          "n_units": n_units, "key_dict": key_stored.dict, "replicated": True},
                 "info": {"rows": batch.nrows, "chunks": int(geom.n_chunks), "chunk_rows": int(geom.chunk_rows),
                          "wg_threads": int(geom.wg_threads), "group_cap": cap, "lds_bytes": int(geom.lds_bytes),
                          "tier": "shared, probe fused"}}
            if world > 1:
                p["gathered"] = torch.zeros(world * table_bytes + PAD, dtype=torch.uint8, device=self.device)
                p["merged"] = torch.zeros(table_bytes + PAD, dtype=torch.uint8, device=self.device)
            if cache_key is not None:
                if len(preps) >= 8:
                    preps.pop(next(iter(preps)))
                preps[key] = p
        if self.rec is not None:
            self.rec.keep.append(p)
        xbuf, slots = p["xbuf"], p["slots"]
        keys_ptr, acc_ptr = xbuf.data_ptr() + 16, xbuf.data_ptr() + 16 + slots * 8
        hs.check(self.lib.hs_agg_shared_join8(self.stream, p["cols"], p["n_cols"], p["key_slot"], p["unit_slot"],
                                              C.byref(p["join"]), p["n_units"], C.byref(p["prog"]), C.byref(p["spec"]),
                                              p["d_units"].data_ptr(), C.byref(p["geom"]), p["out_rep"].data_ptr(), keys_ptr,
                                              acc_ptr, p["ws"].data_ptr(), self.flags.data_ptr(), self._event_handle(0),
                                              self._event_handle(1)), "hs_agg_shared_join8")
        self.last_scan = p["info"]
        self.last_group_cap = cap
        if raw_tables is not None:
            # a streamed probe side (the engine's _run_join_stage_streamed): this range's raw per-JoinJob tables are set
            # aside; join8_finish_ranges adds the ranges up BEFORE the one rounding per JoinJob, like ranks are added up
            nb = p["table_bytes"]
            copy = self.empty(nb, torch.uint8)
            self.op(copy.copy_, xbuf[:nb])
            raw_tables.append(copy)
            return p
        if world > 1:
            from .distributed import all_gather_into  # noqa: PLC0415

            dist, group, _ = dist_ctx
            nb = p["table_bytes"]
            self.op(xbuf[:4].view(torch.int32).copy_, self.flags[:1])  # this rank's status travels in the header
            timed = self.exchange_events
            if timed is not None:
                self.op(timed[2].record)
            self.op(all_gather_into, dist, xbuf[:nb], p["gathered"][: world * nb], group)
            if timed is not None:
                self.op(timed[3].record)
            merged = p["merged"]
            keys_ptr, acc_ptr = merged.data_ptr() + 16, merged.data_ptr() + 16 + slots * 8
            hs.check(self.lib.hs_agg_units_merge(self.stream, p["gathered"].data_ptr(), world, p["n_units"], p["unit_cap"],
                                                 C.byref(p["spec"]), keys_ptr, acc_ptr, self.flags.data_ptr()),
                     "hs_agg_units_merge")
        return self._join8_tables_to_batch(p, keys_ptr, acc_ptr, out_schema)

    def _join8_tables_to_batch(self, p: dict, keys_ptr: int, acc_ptr: int, out_schema: Schema) -> DBatch:
        """Raw per-JoinJob tables -> rounded like a shuffle-file write into the exchange slab -> the short tail's batch."""
        slots = p["slots"]
        hs.check(self.lib.hs_agg_units_to_slab(self.stream, keys_ptr, acc_ptr, p["n_units"], p["unit_cap"], C.byref(p["spec"]),
                                               p["slab"].data_ptr(), C.byref(p["desc"]), self.flags.data_ptr()),
                 "hs_agg_units_to_slab")
        return DBatch(list(out_schema), [], slots, [0, slots], None, total_units=p["n_units"], slab=p["slab"],
                      slab_layout=p["layout"], tail=p["tail"])

    def join8_finish_ranges(self, p: dict, raw_tables: Sequence[torch.Tensor], out_schema: Schema) -> DBatch:
        """The raw unit tables of a streamed probe side's ranges, added up in range order (hs_agg_units_merge: the same
        launch that adds up ranks) BEFORE the one rounding the reference applies per JoinJob (tasks.py:373 -> io.py:87-94),
        then the short tail's batch.  Every range ran with the same capacities (same query, same hints)."""
        nb, slots, n = p["table_bytes"], p["slots"], len(raw_tables)
        gathered = self.empty(n * nb, torch.uint8)
        for r, t in enumerate(raw_tables):
            self.op(gathered[r * nb: (r + 1) * nb].copy_, t[:nb])
        merged = torch.zeros(nb + PAD, dtype=torch.uint8, device=self.device)
        keys_ptr, acc_ptr = merged.data_ptr() + 16, merged.data_ptr() + 16 + slots * 8
        hs.check(self.lib.hs_agg_units_merge(self.stream, gathered.data_ptr(), n, p["n_units"], p["unit_cap"], C.byref(p["spec"]),
                                             keys_ptr, acc_ptr, self.flags.data_ptr()), "hs_agg_units_merge")
        batch = self._join8_tables_to_batch(p, keys_ptr, acc_ptr, out_schema)
        batch.keep = (gathered, merged)
        return batch

    # ---- hash join (A8) --------------------------------------------------------------------------------
    JOIN_DENSE_SPREAD = 32  # slots per build row at most (TPC-H order keys use 8 of every 32 values) ...
    JOIN_DENSE_CROWD = 4    # ... and build rows per slot at most on average: a few hot keys would leave the assembly to single waves

    def _join_indices_dense(self, left_key: DCol, right_key: DCol) -> tuple | None:
        """INTEGER keys over a dense range (round 4, hs_join_dense_*): the build rows are range-partitioned and assembled
        into a CSR over key slots - no hash table, no global atomics; the probe reads two adjacent offsets per row.  None:
        this shape keeps the hash-table join (other key kinds, a sparse or a crowded key range)."""
        n_left, n_right = left_key.n, right_key.n
        if (left_key.kind != hs.I32 or right_key.kind != hs.I32 or n_left == 0 or n_right == 0 or n_left >= 0xFFFFFFFF
                or left_key.data.data_ptr() % 16 or right_key.data.data_ptr() % 16):
            return None
        lo, hi, _ = self.key_range(left_key)
        slots = hi - lo + 1
        if slots > min(1 << 29, self.JOIN_DENSE_SPREAD * n_left + 65536) or slots * self.JOIN_DENSE_CROWD < n_left or n_left >= 1 << 31:
            return None
        ws_bytes = int(self.lib.hs_join_dense_ws_bytes(n_left, slots))
        if ws_bytes == 0:
            return None
        words = self.empty(slots, torch.int32)
        rows = self.empty(n_left, torch.int32)
        list_count = self.empty(n_left, torch.int32)
        ws = self.workspace(ws_bytes)
        hs.check(self.lib.hs_join_dense_build(self.stream, left_key.data.data_ptr(), n_left, lo, slots, words.data_ptr(),
                                              rows.data_ptr(), list_count.data_ptr(), ws.data_ptr(), self.flags.data_ptr()),
                 "hs_join_dense_build")
        counts = self.empty(max(n_right, 1), torch.int64)
        aux = self.workspace(self.lib.hs_join_dense_aux_bytes(n_right))
        hs.check(self.lib.hs_join_dense_count(self.stream, right_key.data.data_ptr(), n_right, lo, slots, words.data_ptr(),
                                              rows.data_ptr(), list_count.data_ptr(), counts.data_ptr(), aux.data_ptr()),
                 "hs_join_dense_count")
        out_start = self.empty(n_right + 1, torch.int64)
        ws2 = self.workspace(self.lib.hs_scan_ws_bytes(n_right))
        hs.check(self.lib.hs_exclusive_scan_i64(self.stream, counts.data_ptr(), n_right, out_start.data_ptr(), ws2.data_ptr()),
                 "hs_exclusive_scan_i64")
        n_out = self.host_int(out_start[n_right])  # sizes the pair lists: the run is data-dependent (not replayable)
        out_left = self.empty(max(n_out, 1), torch.int64)
        out_right = self.empty(max(n_out, 1), torch.int64)
        hs.check(self.lib.hs_join_dense_fill(self.stream, n_right, rows.data_ptr(), aux.data_ptr(), out_start.data_ptr(),
                                             out_left.data_ptr(), out_right.data_ptr()), "hs_join_dense_fill")
        self.last_join = {"mode": "dense csr", "slots": slots, "n_build": n_left}
        self.dense_joins = getattr(self, "dense_joins", 0) + 1
        return out_left, out_right, out_start, n_out

    def _join_indices_hashed(self, left_key: DCol, right_key: DCol) -> tuple | None:
        """Any INTEGER keys (round 4, hs_join_hash_*): the build rows are moved into the order of their hash windows and every
        window of the {key, word} table is assembled in LDS - no global atomics; the probe reads one 8-byte slot per row in
        the usual case and shares the dense form's second pass.  None: other key kinds, more than ~38 M build rows, or a
        window that overflowed (a degenerate key set) - the hash table in global memory takes those."""
        n_left, n_right = left_key.n, right_key.n
        if (left_key.kind != hs.I32 or right_key.kind != hs.I32 or n_left == 0 or n_right == 0
                or left_key.data.data_ptr() % 16 or right_key.data.data_ptr() % 16):
            return None
        slots = int(self.lib.hs_join_hash_slots(n_left))
        if slots == 0:
            return None
        table = self.empty(slots, torch.int64)
        rows = self.empty(n_left, torch.int32)
        list_count = self.empty(n_left, torch.int32)
        ws = self.workspace(int(self.lib.hs_join_hash_ws_bytes(n_left)))
        overflowed = self.empty(1, torch.int32)  # a status word of this build's own: "a window overflowed" is not an error
        overflowed.zero_()
        hs.check(self.lib.hs_join_hash_build(self.stream, left_key.data.data_ptr(), n_left, table.data_ptr(), rows.data_ptr(),
                                             list_count.data_ptr(), ws.data_ptr(), overflowed.data_ptr()), "hs_join_hash_build")
        counts = self.empty(max(n_right, 1), torch.int64)
        aux = self.workspace(self.lib.hs_join_dense_aux_bytes(n_right))
        hs.check(self.lib.hs_join_hash_count(self.stream, right_key.data.data_ptr(), n_right, n_left, table.data_ptr(),
                                             rows.data_ptr(), list_count.data_ptr(), counts.data_ptr(), aux.data_ptr()),
                 "hs_join_hash_count")
        out_start = self.empty(n_right + 1, torch.int64)
        ws2 = self.workspace(self.lib.hs_scan_ws_bytes(n_right))
        hs.check(self.lib.hs_exclusive_scan_i64(self.stream, counts.data_ptr(), n_right, out_start.data_ptr(), ws2.data_ptr()),
                 "hs_exclusive_scan_i64")
        n_out = self.host_int(out_start[n_right])  # sizes the pair lists: the run is data-dependent (not replayable)
        if int(overflowed.item()) != 0:  # (the stream is idle after the read above: this one costs no second wait)
            return None
        out_left = self.empty(max(n_out, 1), torch.int64)
        out_right = self.empty(max(n_out, 1), torch.int64)
        hs.check(self.lib.hs_join_dense_fill(self.stream, n_right, rows.data_ptr(), aux.data_ptr(), out_start.data_ptr(),
                                             out_left.data_ptr(), out_right.data_ptr()), "hs_join_dense_fill")
        self.last_join = {"mode": "hashed windows", "slots": slots, "n_build": n_left}
        self.hashed_joins = getattr(self, "hashed_joins", 0) + 1
        return out_left, out_right, out_start, n_out

    def join_indices(self, left_key: DCol, right_key: DCol) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, int]:
        """Inner equi-join -> (left_rows, right_rows, out_start, n_out): pairs ordered by right row,
        then by left row (the reference's emission order, tasks.py:224-240)."""
        n_left, n_right = left_key.n, right_key.n
        dense = self._join_indices_dense(left_key, right_key) or self._join_indices_hashed(left_key, right_key)
        if dense is not None:
            return dense
        self.last_join = {"mode": "global hash table", "n_build": n_left}
        cap = 16
        while cap < 2 * max(n_left, 1):
            cap *= 2
        tkeys = self.empty(cap, torch.int64)
        treps = self.empty(cap, torch.int64)
        slot_start = self.empty(cap + 1, torch.int64)
        rows = self.empty(max(n_left, 1), torch.int64)
        ws = self.workspace(self.lib.hs_join_build_ws_bytes(n_left, cap))
        lk, rk = left_key.as_hs(), right_key.as_hs()
        hs.check(self.lib.hs_join_build(self.stream, C.byref(lk), n_left, cap, tkeys.data_ptr(), treps.data_ptr(),
                                        slot_start.data_ptr(), rows.data_ptr(), ws.data_ptr(), self.flags.data_ptr()),
                 "hs_join_build")
        counts = self.empty(max(n_right, 1), torch.int64)
        hs.check(self.lib.hs_join_count(self.stream, C.byref(lk), C.byref(rk), n_right, cap, tkeys.data_ptr(),
                                        treps.data_ptr(), slot_start.data_ptr(), counts.data_ptr()), "hs_join_count")
        out_start = self.empty(n_right + 1, torch.int64)
        ws2 = self.workspace(self.lib.hs_scan_ws_bytes(n_right))
        hs.check(self.lib.hs_exclusive_scan_i64(self.stream, counts.data_ptr(), n_right, out_start.data_ptr(),
                                                ws2.data_ptr()), "hs_exclusive_scan_i64")
        n_out = self.host_int(out_start[n_right])  # sizes the pair lists: the run is data-dependent (not replayable)
        out_left = self.empty(max(n_out, 1), torch.int64)
        out_right = self.empty(max(n_out, 1), torch.int64)
        hs.check(self.lib.hs_join_fill(self.stream, C.byref(lk), C.byref(rk), n_right, cap, tkeys.data_ptr(),
                                       treps.data_ptr(), slot_start.data_ptr(), rows.data_ptr(), out_start.data_ptr(),
                                       out_left.data_ptr(), out_right.data_ptr()), "hs_join_fill")
        return out_left, out_right, out_start, n_out
