"""Multi-GPU exchange of partial-aggregate rows (the shuffle step between the two aggregation phases).

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).  File block ``b`` lives on
rank ``b % world`` (reference: one ScanJob per block, plan.py:90-93 - independent units), so the scan +
partial aggregate needs no communication.  The reference then routes every partial row through its
shuffle files to the final merge (tasks.py:347-375, plan.py:94-98); here the partial rows of all ranks
meet through ONE collective per query:

* every rank packs its partial rows into a fixed-size *slab*: header (status flags, row count), the
  global unit id (= file block id) of every row, then the columns, each padded to ``slab_rows`` rows;
  the producing kernels write straight into the slab, there is no packing pass;
* ``all_gather_into_tensor`` of the slabs (a few KB per rank: latency-bound, one RCCL call);
* every rank un-interleaves the columns (one launch of hs_slab_unpack; ``unpack_gathered`` below is its torch
  statement, used by the CPU tests) and runs the same final merge, which folds the partials of a key
  in ascending (block id, row) order - exactly the order in which the reference's single shuffle file
  holds them - so the result does not depend on the number of GPUs.  Rank 0 writes the result file.

Fixed-size slabs mean no count exchange and no host round trip before the collective.  This module is
plain torch tensor plumbing and device-agnostic: tests/test_distributed_cpu.py runs it under gloo.
"""

from __future__ import annotations

import os

from dataclasses import dataclass, field
from typing import Any

import torch

HEADER_BYTES = 16  # [flags: u32][pad: u32][row count: i64]
FLAG_BITS = 32     # width of the status word or_flags() reduces (every HS_FLAG_* bit)


@dataclass
class SlabColumn:
    offset: int  # byte offset inside the slab
    row_bytes: int  # bytes per row
    dtype: torch.dtype  # element type of the column view (uint8 for packed strings)


@dataclass
class SlabLayout:
    """Byte layout of one rank's slab: header | order keys (i64 per row) | columns."""

    slab_rows: int
    columns: list[SlabColumn] = field(default_factory=list)
    order_offset: int = HEADER_BYTES
    nbytes: int = 0

    @staticmethod
    def build(slab_rows: int, column_specs: list[tuple[int, torch.dtype]]) -> "SlabLayout":
        """column_specs: (bytes per row, view dtype) per column."""
        lay = SlabLayout(slab_rows)
        pos = HEADER_BYTES + 8 * slab_rows
        for row_bytes, dtype in column_specs:
            pos = (pos + 15) & ~15
            lay.columns.append(SlabColumn(pos, row_bytes, dtype))
            pos += row_bytes * slab_rows
        lay.nbytes = (pos + 15) & ~15
        return lay

    # ---- views into a rank's slab --------------------------------------------------------------------
    def order_view(self, slab: torch.Tensor) -> torch.Tensor:
        return slab[self.order_offset: self.order_offset + 8 * self.slab_rows].view(torch.int64)

    def column_view(self, slab: torch.Tensor, i: int) -> torch.Tensor:
        c = self.columns[i]
        return slab[c.offset: c.offset + c.row_bytes * self.slab_rows].view(c.dtype)

    def flags_view(self, slab: torch.Tensor) -> torch.Tensor:
        return slab[0:4].view(torch.int32)

    def count_view(self, slab: torch.Tensor) -> torch.Tensor:
        return slab[8:16].view(torch.int64)


def all_gather_slabs(dist: Any, slab: torch.Tensor, world: int, group: Any = None) -> torch.Tensor:
    """-> uint8 tensor [world, slab_bytes].  RCCL moves device tensors directly; the gloo backend (CPU
    tests, single-GPU rehearsal) has no device all-gather, so device tensors are staged through the host."""
    out = torch.empty(world * slab.numel(), dtype=torch.uint8, device=slab.device)
    backend = dist.get_backend(group)
    if backend == "gloo" and slab.device.type != "cpu":
        host_in = slab.cpu()
        host_out = torch.empty(world * slab.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(host_out, host_in, group=group)
        out.copy_(host_out)
    else:
        dist.all_gather_into_tensor(out, slab, group=group)
    return out.view(world, slab.numel())


def all_gather_slabs_into(dist: Any, slab: torch.Tensor, out: torch.Tensor, group: Any = None) -> None:
    """all_gather_slabs writing into a caller-owned buffer ``out`` (uint8, world * slab bytes): the form the
    engine records and replays."""
    backend = dist.get_backend(group)
    if backend == "gloo" and slab.device.type != "cpu":
        host_out = torch.empty(out.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(host_out, slab.cpu(), group=group)
        out.copy_(host_out)
    else:
        dist.all_gather_into_tensor(out, slab, group=group)


def all_gather_into(dist: Any, inp: torch.Tensor, out: torch.Tensor, group: Any = None) -> None:
    """all_gather_into_tensor of any dtype into a caller-owned buffer (out.numel() == world * inp.numel()); device
    tensors are staged through the host under gloo (CPU tests, single-GPU rehearsal)."""
    backend = dist.get_backend(group)
    if backend == "gloo" and inp.device.type != "cpu":
        host_out = torch.empty(out.numel(), dtype=out.dtype)
        dist.all_gather_into_tensor(host_out, inp.cpu(), group=group)
        out.copy_(host_out)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def unpack_gathered(gathered: torch.Tensor, layout: SlabLayout) -> tuple[torch.Tensor, torch.Tensor, list[torch.Tensor]]:
    """[world, slab_bytes] -> (flags int32[world], order int64[world*M] with -1 on padding rows,
    columns, each contiguous over all world*M rows in rank-major order)."""
    world = gathered.shape[0]
    m = layout.slab_rows
    flags = gathered[:, 0:4].contiguous().view(torch.int32).reshape(world)
    counts = gathered[:, 8:16].contiguous().view(torch.int64).reshape(world)
    order = gathered[:, layout.order_offset: layout.order_offset + 8 * m].contiguous().view(torch.int64).reshape(world, m)
    # rows at or beyond a rank's count are padding: their order key must be negative
    valid = torch.arange(m, device=gathered.device).unsqueeze(0) < counts.unsqueeze(1)
    order = torch.where(valid, order, torch.full_like(order, -1)).reshape(world * m)
    cols = []
    for c in layout.columns:
        cols.append(gathered[:, c.offset: c.offset + c.row_bytes * m].contiguous().view(c.dtype).reshape(-1))
    return flags, order, cols


def local_blocks(n_blocks: int, rank: int, world: int) -> list[int]:
    return [b for b in range(n_blocks) if b % world == rank]


def max_local_units(n_blocks: int, world: int) -> int:
    return (n_blocks + world - 1) // world


# ---- generic row exchange: all-to-all-v by destination rank ---------------------------------------------------
def _a2a(dist: Any, out: torch.Tensor, inp: torch.Tensor, out_splits: list[int], in_splits: list[int],
         group: Any = None) -> None:
    """all_to_all_single with split sizes; device tensors are staged through the host under gloo."""
    backend = dist.get_backend(group)
    if backend == "gloo" and inp.device.type != "cpu":
        host_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host_out, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(host_out)
    else:
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)


def exchange_counts(dist: Any, send_counts: list[int], device: torch.device, group: Any = None) -> list[int]:
    """Tell every rank how many items it will receive from each rank (one tiny all-to-all + D2H)."""
    world = len(send_counts)
    backend = dist.get_backend(group)
    dev = torch.device("cpu") if backend == "gloo" else device
    inp = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    out = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(out, inp, group=group)
    return [int(v) for v in out.tolist()]


def exchange_size_matrix(dist: Any, mine: list[list[int]], device: torch.device, group: Any = None) -> list[list[int]]:
    """mine[d] = the k sizes this rank sends to rank d (rows, payload bytes ...); -> theirs[s] = the k sizes rank s
    sends here.  One tiny all-to-all (+ the read-back the split sizes of the data collective need anyway)."""
    world, k = len(mine), len(mine[0]) if mine else 0
    backend = dist.get_backend(group)
    dev = torch.device("cpu") if backend == "gloo" else device
    inp = torch.tensor(mine, dtype=torch.int64, device=dev).reshape(world * k)
    out = torch.empty(world * k, dtype=torch.int64, device=dev)
    dist.all_to_all_single(out, inp, group=group)
    flat = [int(v) for v in out.tolist()]
    return [flat[s * k: (s + 1) * k] for s in range(world)]


def all_to_all_rows(dist: Any, send: torch.Tensor, send_counts: list[int], recv_counts: list[int], elems_per_row: int = 1,
                    group: Any = None) -> torch.Tensor:
    """Rows of ``send`` are grouped by destination rank (send_counts rows each); returns the rows received,
    grouped by source rank.  RCCL runs this as direct peer-to-peer transfers over the xGMI mesh."""
    n_out = sum(recv_counts) * elems_per_row
    out = torch.empty(n_out + 16, dtype=send.dtype, device=send.device)[:n_out]
    _a2a(dist, out, send[: sum(send_counts) * elems_per_row].contiguous(), [c * elems_per_row for c in recv_counts],
         [c * elems_per_row for c in send_counts], group)
    return out


def agree_string_width(dist: Any, fixed_len: int, nrows: int, device: torch.device, group: Any = None) -> int:
    """The byte length shared by the strings of a column over ALL ranks, or -1 (variable).  A rank's own view
    (``fixed_len`` from the min / max of its local rows, 0 rows -> no opinion) must not decide between exchange
    forms: a rank that owns no block, or whose blocks happen to hold equal-length keys while another's do not, would
    pick another collective than its peers and the job would hang.  One tiny all-reduce; every rank gets the same
    answer."""
    if nrows == 0:
        lo, hi = 1 << 20, -1      # no opinion
    elif fixed_len >= 0:
        lo = hi = int(fixed_len)
    else:
        lo, hi = 0, 1 << 20       # lengths differ already on this rank
    backend = dist.get_backend(group)
    dev = torch.device("cpu") if backend == "gloo" else device
    t = torch.tensor([-lo, hi], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    neg_lo, hi = (int(v) for v in t.tolist())
    if hi < 0:
        return 0  # the column is empty everywhere
    return -neg_lo if -neg_lo == hi else -1


def or_flags(dist: Any, flags: int, device: torch.device, group: Any = None) -> int:
    """Bitwise OR of a 32-bit status word over all ranks (so every rank takes the same retry / error decision).
    RCCL has no BOR reduction: the word travels as one element per bit under MAX.  All 32 bits - round 2 reduced only
    the low byte and dropped HS_FLAG_MERGE_FULL / HS_FLAG_MERGE_ROWS (bits 8-9), see tests/test_distributed_cpu.py."""
    backend = dist.get_backend(group)
    dev = torch.device("cpu") if backend == "gloo" else device
    flags = int(flags) & 0xFFFFFFFF
    bits = torch.tensor([(flags >> b) & 1 for b in range(FLAG_BITS)], dtype=torch.int32, device=dev)
    dist.all_reduce(bits, op=dist.ReduceOp.MAX, group=group)
    return sum(int(v) << b for b, v in enumerate(bits.tolist()))


class PeerSlabsUnavailable(RuntimeError):
    """The peers' buffers cannot be mapped on some rank (raised on EVERY rank together): the caller keeps the all-gather."""


class PeerSlabs:
    """Peer-to-peer exchange of the short tail's slabs (prototype, HIPSPARK_P2P_SLABS=1; csrc/hs_exchange.hip hs_slab_push /
    hs_slab_wait): every rank owns one device buffer that all peers map through hipIpc handles (exchanged ONCE here, over the
    process group's object collective); per query a rank stores its slab into every peer's buffer and raises a flag there,
    and waits on the device for the flags of all ranks - no collective, no host step in the data path.  Replaces
    ``all_gather_slabs_into`` for slabs of at most ``SLOT`` bytes; on one node only (hipIpc); tested at world 1 over RCCL
    and with 2 - 3 processes sharing the test box's GPU - the 8-GPU xGMI case has not run anywhere."""

    SLOT = 256 * 1024
    # bound of the device-side wait (HS_FLAG_PEER_TIMEOUT).  The decision is RANK-LOCAL by design - reducing the status word
    # over the ranks would put back the collective this form removes: a rank whose wait ran out raises DeviceError while its
    # peers may still return that query's result; they run out on their next query (the raising rank pushes no more).  A
    # timeout therefore means "a peer is lost", never "a peer is slow": runs that are not yet recorded (a peer may still be
    # in its first-run hiprtc compile) wait ten times as long.
    TIMEOUT_MS = int(os.environ.get("HIPSPARK_P2P_TIMEOUT_MS", "20000"))

    class _Handle(__import__("ctypes").Structure):
        _fields_ = [("reserved", __import__("ctypes").c_char * 64)]

    def __init__(self, dist: Any, group: Any, rank: int, world: int, device: Any, lib: Any) -> None:
        import ctypes as C  # noqa: PLC0415

        self.lib, self.rank, self.world, self.device = lib, rank, world, device
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), PeerSlabs._Handle, C.c_uint]
        self.nbytes = int(lib.hs_slab_p2p_bytes(world, self.SLOT))
        with torch.cuda.device(device):
            own = C.c_void_p()
            # fine-grained device memory: a peer's stores and flag writes arrive over the fabric, not through this GPU's L2 -
            # the waiting kernel's system-scope loads must see them while it runs (what RCCL does for its own flags)
            self.fine_grained = self.hip.hipExtMallocWithFlags(C.byref(own), C.c_size_t(self.nbytes), C.c_uint(0x1)) == 0
            problem = None
            if not self.fine_grained:
                # a coarse-grained buffer would let k_slab_wait spin on a flag its L2 never refreshes, or copy stale slab bytes:
                # across ranks that is a `problem` like a failed mapping (every rank then keeps the all-gather together); at
                # world 1 the only writer is this GPU itself and ordinary device memory is sound
                own = C.c_void_p()
                if world > 1:
                    problem = "hipExtMallocWithFlags(fine-grained) failed: no peer-visible buffer on this rank"
                else:
                    self._check(self.hip.hipMalloc(C.byref(own), C.c_size_t(self.nbytes)), "hipMalloc")
            if own.value:
                self._check(self.hip.hipMemset(own, 0, C.c_size_t(self.nbytes)), "hipMemset")
                self._check(self.hip.hipDeviceSynchronize(), "hipDeviceSynchronize")
            self.own = own.value or 0
            self.opened: list[int] = []
            peers = [self.own] * world
            if world > 1:
                handle = PeerSlabs._Handle()
                sent = b""  # a rank without a buffer sends a handle its peers reject: all of them then report a problem
                if problem is None:
                    self._check(self.hip.hipIpcGetMemHandle(C.byref(handle), own), "hipIpcGetMemHandle")
                    sent = C.string_at(C.byref(handle), 64)  # all 64 bytes (they contain NULs)
                handles: list = [None] * world
                dist.all_gather_object(handles, sent, group=group)
                for p in range(world):
                    if p == rank or problem is not None:
                        continue
                    h = PeerSlabs._Handle()
                    if len(handles[p]) != 64:
                        problem = f"rank {p} has no mappable buffer (IPC handle of {len(handles[p])} bytes)"
                        break
                    C.memmove(C.byref(h), handles[p], 64)
                    ptr = C.c_void_p()
                    rc = self.hip.hipIpcOpenMemHandle(C.byref(ptr), h, 1)  # 1: lazy peer access
                    if rc != 0:
                        problem = f"hipIpcOpenMemHandle of rank {p}'s buffer failed with hip error {rc}"
                        break
                    peers[p] = ptr.value
                    self.opened.append(ptr.value)
                # the mapping either works on EVERY rank or the exchange form is not used at all: a rank that raised alone
                # would leave its peers waiting in the next collective
                problems: list = [None] * world
                dist.all_gather_object(problems, problem, group=group)
                if any(problems):
                    self.close()
                    raise PeerSlabsUnavailable("; ".join(f"rank {r}: {m}" for r, m in enumerate(problems) if m))
            self.peers_dev = torch.tensor(peers, dtype=torch.int64, device=device)
            self.epochs = torch.zeros(2, dtype=torch.int64, device=device)
        if world > 1:
            dist.barrier(group=group)  # every buffer is zeroed and mapped before the first push

    @staticmethod
    def _check(rc: int, what: str) -> None:
        if rc != 0:
            raise RuntimeError(f"peer-to-peer slabs: {what} failed with hip error {rc}")

    def fits(self, slab_bytes: int) -> bool:
        return 16 <= slab_bytes <= self.SLOT

    def push(self, stream: Any, slab: torch.Tensor) -> None:
        from . import hipspark as hs  # noqa: PLC0415

        hs.check(self.lib.hs_slab_push(stream, slab.data_ptr(), slab.numel(), self.peers_dev.data_ptr(), self.world, self.rank,
                                       self.SLOT, self.epochs.data_ptr()), "hs_slab_push")

    def wait_into(self, stream: Any, slab_bytes: int, gathered: torch.Tensor, flags_ptr: int, recorded: bool = True) -> None:
        from . import hipspark as hs  # noqa: PLC0415

        hs.check(self.lib.hs_slab_wait(stream, self.own, self.world, self.SLOT, slab_bytes, self.epochs.data_ptr(),
                                       gathered.data_ptr(), slab_bytes, flags_ptr, self.TIMEOUT_MS * (1 if recorded else 10)),
                 "hs_slab_wait")

    def close(self) -> None:
        import ctypes as C  # noqa: PLC0415

        for ptr in self.opened:
            self.hip.hipIpcCloseMemHandle(C.c_void_p(ptr))
        self.opened = []
        if self.own:
            self.hip.hipFree(C.c_void_p(self.own))
            self.own = 0
