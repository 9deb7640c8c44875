"""ORACLE (test infrastructure): ctypes access to oracle/q45_oracle.c - the plain-C restatements of the reference's
path for BASELINE configs 4 (join + GROUP BY) and 5 (LIKE + CONCAT-key GROUP BY).  Built with gcc on first use (no
fast-math, no FMA contraction).  Only tests/, __graft_entry__ and bench.py's checker / cpu_baseline legs import this."""

from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SRC = HERE / "q45_oracle.c"
LIB = HERE / "_build" / "libq45oracle.so"
CFLAGS = ["-O2", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-std=gnu11"]


class q4_row(C.Structure):
    _fields_ = [("code", C.c_int32), ("n", C.c_int32), ("qty", C.c_double), ("revenue", C.c_double),
                ("max_price", C.c_double)]


class q5_row(C.Structure):
    _fields_ = [("flag", C.c_int32), ("mode", C.c_int32), ("count", C.c_int32), ("pad", C.c_int32),
                ("qty", C.c_double), ("avg_disc", C.c_double)]


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
        _lib.q4_run.restype = C.c_int
        _lib.q5_run.restype = C.c_int
        _lib.q4_gen_orders.restype = None
        _lib.q4_orders_multiplier.restype = C.c_uint64
    return _lib


def _p(a):
    return None if a is None else np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)


def encode(strings: list[str]) -> tuple[np.ndarray, list[str]]:
    """Dictionary-code a Python string column: (u8 codes, dictionary in first-appearance order)."""
    table: dict[str, int] = {}
    codes = np.empty(len(strings), np.uint8)
    for i, s in enumerate(strings):
        codes[i] = table.setdefault(s, len(table))
    if len(table) > 256:
        raise ValueError("more than 256 distinct strings")
    return codes, list(table)


def run_join_group(o_key: np.ndarray, o_code: np.ndarray, priorities: list[str], l_key: np.ndarray, l_qty: np.ndarray,
                   l_price: np.ndarray, threads: int = 1) -> list[dict]:
    """Rows of workloads.join_group (config 4): o_orderpriority, n, qty, revenue, max_price."""
    o_key, o_code = np.ascontiguousarray(o_key, np.int32), np.ascontiguousarray(o_code, np.uint8)
    l_key = np.ascontiguousarray(l_key, np.int32)
    l_qty, l_price = np.ascontiguousarray(l_qty, np.float32), np.ascontiguousarray(l_price, np.float32)
    rows = (q4_row * 256)()
    n = lib().q4_run(_p(o_key), _p(o_code), C.c_int64(len(o_key)), _p(l_key), _p(l_qty), _p(l_price),
                     C.c_int64(len(l_key)), C.c_int32(threads), rows)
    if n == -1:
        raise OverflowError("int too big to convert")
    if n == -3:
        raise AssertionError("FLOAT column holds int")
    return [{"o_orderpriority": priorities[rows[i].code], "n": int(rows[i].n), "qty": rows[i].qty,
             "revenue": rows[i].revenue, "max_price": rows[i].max_price} for i in range(n)]


def run_strkey_like(flag: np.ndarray, mode: np.ndarray, modes: list[str], qty: np.ndarray, disc: np.ndarray,
                    block_rows: list[int], pattern: str = "%AIR%", threads: int = 1) -> list[dict]:
    """Rows of workloads.strkey_like (config 5): k, qty, avg_disc, count."""
    flag, mode = np.ascontiguousarray(flag, np.uint8), np.ascontiguousarray(mode, np.uint8)
    qty, disc = np.ascontiguousarray(qty, np.float32), np.ascontiguousarray(disc, np.float32)
    blob = "".join(modes).encode()
    off = np.zeros(len(modes) + 1, np.int32)
    off[1:] = np.cumsum([len(m.encode()) for m in modes])
    br = np.asarray(block_rows, np.int64)
    pat = np.frombuffer(pattern.encode(), np.uint8)
    rows = (q5_row * 2048)()
    n = lib().q5_run(_p(flag), _p(mode), _p(qty), _p(disc), _p(br), C.c_int32(len(br)), _p(np.frombuffer(blob, np.uint8)),
                     _p(off), C.c_int32(len(modes)), _p(pat), C.c_int32(len(pat)), C.c_int32(threads), rows)
    if n == -1:
        raise OverflowError("int too big to convert")
    if n < 0:
        raise ValueError("q5_run: bad arguments")
    return [{"k": f"{chr(rows[i].flag)}-{modes[rows[i].mode]}", "qty": rows[i].qty, "avg_disc": rows[i].avg_disc,
             "count": int(rows[i].count)} for i in range(n)]


def gen_orders(seed: int, row0: int, n_rows: int, n_total: int) -> tuple[np.ndarray, np.ndarray]:
    """CPU twin of hs_gen_orders: (o_orderkey i32, priority code u8) for rows [row0, row0 + n_rows) of n_total."""
    okey, code = np.empty(n_rows, np.int32), np.empty(n_rows, np.uint8)
    lib().q4_gen_orders(C.c_uint64(seed), C.c_int64(row0), C.c_int64(n_rows), C.c_int64(n_total), _p(okey), _p(code))
    return okey, code
