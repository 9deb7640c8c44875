"""ORACLE (test infrastructure): ctypes access to oracle/q1_oracle.c - the plain-C restatement of the
reference's Q1 path.  Built with gcc on first use (no fast-math, no FMA contraction: the arithmetic
must round exactly like CPython's float).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this."""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SRC = HERE / "q1_oracle.c"
LIB = HERE / "_build" / "libq1oracle.so"
CFLAGS = ["-O2", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-std=c11"]


class q1_row(C.Structure):
    _fields_ = [
        ("key", C.c_int32),
        ("count_order", C.c_int32),
        ("sum_qty", C.c_double),
        ("sum_base_price", C.c_double),
        ("sum_disc_price", C.c_double),
        ("sum_charge", C.c_double),
        ("avg_qty", C.c_double),
        ("avg_price", C.c_double),
        ("avg_disc", C.c_double),
        ("raw", C.c_double * 7),
    ]


_lib = None


def build(force: bool = False) -> Path:
    if force or not LIB.exists() or LIB.stat().st_mtime < SRC.stat().st_mtime:
        LIB.parent.mkdir(exist_ok=True)
        subprocess.run(["gcc", *CFLAGS, "-o", str(LIB), str(SRC), "-lm"], check=True)
    return LIB


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(build()))
        _lib.q1_run_threads.restype = C.c_int
        _lib.q1_gen.restype = None
    return _lib


def _p(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def gen(seed: int, row0: int, n: int, orderkey: bool = False, shipmode: bool = False) -> dict[str, np.ndarray]:
    """CPU twin of hs_gen_lineitem: the synthetic lineitem columns for rows [row0, row0+n)."""
    out = {
        "l_quantity": np.empty(n, np.float32), "l_extendedprice": np.empty(n, np.float32),
        "l_discount": np.empty(n, np.float32), "l_tax": np.empty(n, np.float32),
        "l_shipdate": np.empty(n, np.int64), "l_returnflag": np.empty(n, np.uint8),
    }
    ok = np.empty(n, np.int32) if orderkey else None
    sm = np.empty(n, np.uint8) if shipmode else None
    lib().q1_gen(C.c_uint64(seed), C.c_int64(row0), C.c_int64(n), _p(out["l_quantity"]), _p(out["l_extendedprice"]),
                 _p(out["l_discount"]), _p(out["l_tax"]), _p(out["l_shipdate"]), _p(out["l_returnflag"]), _p(ok), _p(sm))
    if ok is not None:
        out["l_orderkey"] = ok
    if sm is not None:
        out["l_shipmode_code"] = sm
    return out


OUT_COLUMNS = ["sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price", "avg_disc"]


def run(cols: dict[str, np.ndarray], block_rows: list[int], cutoff_us: int, threads: int = 1,
        raw: bool = False) -> list[dict]:
    """Q1 result rows (dict per group, columns named like tests/queries.q1) from columnar inputs."""
    rows = (q1_row * 256)()
    br = np.asarray(block_rows, dtype=np.int64)
    arrs = [np.ascontiguousarray(cols[k]) for k in
            ("l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_shipdate", "l_returnflag")]
    n = lib().q1_run_threads(*[_p(a) for a in arrs], _p(br), C.c_int32(len(br)), C.c_int64(cutoff_us),
                             C.c_int32(threads), rows)
    if n < 0:
        raise OverflowError("int too big to convert")
    return _rows_out(rows, n, raw)


def _rows_out(rows, n: int, raw: bool = False) -> list[dict]:
    out = []
    for i in range(n):
        r = rows[i]
        row = {"l_returnflag": chr(r.key)}
        for j, name in enumerate(OUT_COLUMNS):
            row[name] = r.raw[j] if raw else getattr(r, name)
        row["count_order"] = int(r.count_order)
        out.append(row)
    return out


def run_synth(seed: int, total_rows: int, rows_per_block: int, cutoff_us: int, threads: int = 1) -> list[dict]:
    """Q1 over the synthetic lineitem of ``total_rows`` rows, generated block by block inside the C oracle."""
    rows = (q1_row * 256)()
    lib().q1_run_synth.restype = C.c_int
    n = lib().q1_run_synth(C.c_uint64(seed), C.c_int64(total_rows), C.c_int64(rows_per_block), C.c_int64(cutoff_us),
                           C.c_int32(threads), rows)
    if n == -1:
        raise OverflowError("int too big to convert")
    if n < 0:
        raise ValueError("q1_run_synth: bad arguments")
    return _rows_out(rows, n)


def host_threads() -> int:
    """Usable host cores: the affinity mask, capped by the cgroup CPU quota when there is one (a GPU box
    exposes all host cores in the mask but grants this container only a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n
