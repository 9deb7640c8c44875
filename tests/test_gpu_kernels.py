"""Kernel-level GPU tests through the C ABI: every generic operator against a numpy / pure-Python statement
of the reference loop it replaces, on seeded inputs including the edge cases the domain has (empty inputs,
ragged tails, duplicates, negative keys, empty strings, dictionary collisions)."""

from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from minispark_amd.device import Device

    return Device(0)


def _rng(seed):
    return np.random.default_rng(seed)


def test_synthetic_generator_matches_cpu_twin(dev):
    """k_gen_lineitem (device) == q1_gen (oracle/q1_oracle.c) for an arbitrary row window."""
    import torch

    from minispark_amd import hipspark as hs
    from oracle import q1_native

    row0, n = 2_097_100, 5000  # straddles a block boundary
    t = {k: dev.empty(n, dt) for k, dt in [("q", torch.float32), ("p", torch.float32), ("d", torch.float32),
                                           ("t", torch.float32), ("s", torch.int64), ("f", torch.uint8),
                                           ("l", torch.uint8), ("o", torch.int32), ("m", torch.uint8)]}
    hs.check(dev.lib.hs_gen_lineitem(dev.stream, 20251003, row0, n, *[t[k].data_ptr() for k in "qpdtsflom"]))
    want = q1_native.gen(20251003, row0, n, orderkey=True, shipmode=True)
    for key, name in [("q", "l_quantity"), ("p", "l_extendedprice"), ("d", "l_discount"), ("t", "l_tax"),
                      ("s", "l_shipdate"), ("f", "l_returnflag"), ("o", "l_orderkey"), ("m", "l_shipmode_code")]:
        assert np.array_equal(t[key].cpu().numpy(), want[name]), name
    assert (t["l"].cpu().numpy() == 1).all()


@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 2047, 2048, 2049, 4095, 4096, 4097, 8193, 100_003])
def test_string_offsets_and_fixed_len(dev, n):
    import torch

    lens = _rng(n).integers(0, 256, n).astype(np.uint8)
    d_lens = dev.to_device(lens, torch.uint8)
    col = dev.string_col(d_lens, dev.empty(int(lens.sum()), torch.uint8), n)
    if n and lens.min() == lens.max():
        assert col.fixed_len == int(lens[0]) and col.offs is None
    elif n:
        assert col.fixed_len == -1
        want = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
        assert np.array_equal(col.offs.cpu().numpy(), want)
    else:
        assert col.fixed_len == 0


@pytest.mark.parametrize("n,p", [(0, 0.5), (1, 1.0), (17, 0.5), (4096, 0.5), (4097, 0.0), (4097, 1.0), (250_000, 0.3),
                                 (1_000_003, 0.9)])
def test_compact_is_stable_selection(dev, n, p):
    import torch

    from minispark_amd import hipspark as hs

    mask = (_rng(7).random(n) < p).astype(np.uint8) * _rng(8).integers(1, 255, n).astype(np.uint8)  # any nonzero keeps
    d_mask = dev.to_device(mask, torch.uint8)
    sel = dev.empty(n, torch.int64)
    count = dev.empty(1, torch.int64)
    ws = dev.workspace(dev.lib.hs_scan_ws_bytes(n))
    hs.check(dev.lib.hs_compact(dev.stream, d_mask.data_ptr(), n, sel.data_ptr(), count.data_ptr(), ws.data_ptr()))
    k = int(count.item())
    assert np.array_equal(sel[:k].cpu().numpy(), np.nonzero(mask)[0])


@pytest.mark.parametrize("shift", [1, 5, 16])
def test_scans_accept_unaligned_inputs(dev, shift):
    """Views that do not start on a 16-byte boundary take the generic kernels: same results."""
    import torch

    from minispark_amd import hipspark as hs

    n = 20_000
    raw = _rng(3).integers(0, 200, n + shift).astype(np.uint8)
    d = dev.to_device(raw, torch.uint8)[shift:]
    lens = raw[shift:]
    offs = dev.empty(n + 1, torch.int64)
    mm = dev.empty(2, torch.int32)
    ws = dev.workspace(dev.lib.hs_scan_ws_bytes(n))
    hs.check(dev.lib.hs_str_offsets(dev.stream, d.data_ptr(), n, offs.data_ptr(), mm.data_ptr(), ws.data_ptr()))
    assert np.array_equal(offs.cpu().numpy(), np.concatenate([[0], np.cumsum(lens.astype(np.int64))]))
    assert mm.cpu().tolist() == [int(lens.min()), int(lens.max())]
    sel = dev.empty(n, torch.int64)
    count = dev.empty(1, torch.int64)
    mask = (lens % 3 == 0).astype(np.uint8)
    dm = dev.to_device(np.concatenate([np.zeros(shift, np.uint8), mask]), torch.uint8)[shift:]
    hs.check(dev.lib.hs_compact(dev.stream, dm.data_ptr(), n, sel.data_ptr(), count.data_ptr(), ws.data_ptr()))
    assert np.array_equal(sel[: int(count.item())].cpu().numpy(), np.nonzero(mask)[0])


@pytest.mark.parametrize("n_parts", [1, 7, 10, 16])
def test_partition_follows_python_hash_and_is_stable(dev, n_parts):
    """hash(int) % P with Python semantics (hash(-1) == -2, floor-mod), reference tasks.py:362."""
    from minispark_amd.constants import ColumnType
    from minispark_amd.device import DBatch

    keys = np.concatenate([_rng(3).integers(-50, 50, 20_000), [-1, -1, 0, 2**31 - 1, -(2**31)]]).astype(np.int32)
    col = dev.upload_raw(keys, ColumnType.INTEGER)
    batch = DBatch([("k", ColumnType.INTEGER)], [col], len(keys))
    perm, start = dev.partition(batch, 0, n_parts)
    perm = perm.cpu().numpy()
    want_part = np.array([hash(int(k)) % n_parts for k in keys])
    assert start[-1] == len(keys) and len(start) == n_parts + 1
    for p in range(n_parts):
        rows = perm[start[p]: start[p + 1]]
        assert np.array_equal(rows, np.nonzero(want_part == p)[0]), f"partition {p} not the stable selection"


def test_partition_strings_are_complete(dev):
    from minispark_amd.constants import ColumnType
    from minispark_amd.device import DBatch
    from minispark_amd.io import StrCol

    words = ["", "a", "AIR", "REG AIR", "x" * 255, "same", "same"] * 300
    col = dev.upload_raw(StrCol.from_strings(words), ColumnType.STRING)
    perm, start = dev.partition(DBatch([("s", ColumnType.STRING)], [col], len(words)), 0, 10)
    perm = perm.cpu().numpy()
    assert sorted(perm.tolist()) == list(range(len(words)))
    part_of = {}
    for p in range(10):
        for r in perm[start[p]: start[p + 1]]:
            assert part_of.setdefault(words[r], p) == p, "equal strings must land in one partition"


def test_join_matches_reference_order(dev):
    """Inner equi-join with duplicate and missing keys: pairs ordered by right row, then left row ascending
    (reference tasks.py:224-240)."""
    from minispark_amd.constants import ColumnType

    left = _rng(1).integers(-20, 20, 3000).astype(np.int32)
    right = _rng(2).integers(-25, 25, 5000).astype(np.int32)
    lcol, rcol = dev.upload_raw(left, ColumnType.INTEGER), dev.upload_raw(right, ColumnType.INTEGER)
    out_left, out_right, _, n_out = dev.join_indices(lcol, rcol)
    rows_of = {}
    for i, k in enumerate(left.tolist()):
        rows_of.setdefault(k, []).append(i)
    want = [(li, ri) for ri, k in enumerate(right.tolist()) for li in rows_of.get(k, [])]
    got = list(zip(out_left[:n_out].cpu().numpy().tolist(), out_right[:n_out].cpu().numpy().tolist()))
    assert got == want


def test_join_on_strings_including_long_ones(dev):
    from minispark_amd.constants import ColumnType
    from minispark_amd.io import StrCol

    lwords = ["apple", "banana", "a-long-key-beyond-seven-bytes", "apple", "", "a-long-key-beyond-seven-byteS"]
    rwords = ["", "apple", "cherry", "a-long-key-beyond-seven-bytes", "banana", "apple"]
    out_left, out_right, _, n = dev.join_indices(dev.upload_raw(StrCol.from_strings(lwords), ColumnType.STRING),
                                                 dev.upload_raw(StrCol.from_strings(rwords), ColumnType.STRING))
    got = list(zip(out_left[:n].cpu().numpy().tolist(), out_right[:n].cpu().numpy().tolist()))
    want = [(li, ri) for ri, w in enumerate(rwords) for li, v in enumerate(lwords) if v == w]
    assert got == want


def test_expression_evaluator_against_python_semantics(dev):
    """hs_eval vs the oracle's row evaluator (Python operators): mixed int/float arithmetic, floor division
    and modulo with negative operands, comparisons, & / |, LIKE, string comparison."""
    from datetime import datetime

    from minispark_amd.constants import ColumnType as T
    from minispark_amd.device import DBatch
    from minispark_amd.io import StrCol
    from minispark_amd.sql import Col, Lit
    from oracle.py_engine import compile_expr

    n = 4000
    r = _rng(11)
    i = r.integers(-1000, 1000, n).astype(np.int32)
    j = np.where(r.random(n) < 0.5, r.integers(1, 17, n), -r.integers(1, 17, n)).astype(np.int32)
    f = (r.integers(-5000, 5000, n) / 8.0).astype(np.float32)
    g = np.where(r.random(n) < 0.5, 1.5, -2.25).astype(np.float32)
    words = [["REG AIR", "AIR", "RAIL", "SHIP", "", "AIRMAIL", "xAIRx"][k] for k in r.integers(0, 7, n)]
    ts = (r.integers(0, 10_000, n).astype(np.int64) * 86_400_000_000)
    schema = [("i", T.INTEGER), ("j", T.INTEGER), ("f", T.FLOAT), ("g", T.FLOAT), ("s", T.STRING), ("t", T.TIMESTAMP)]
    cols = [dev.upload_raw(i, T.INTEGER), dev.upload_raw(j, T.INTEGER), dev.upload_raw(f, T.FLOAT),
            dev.upload_raw(g, T.FLOAT), dev.upload_raw(StrCol.from_strings(words), T.STRING), dev.upload_raw(ts, T.TIMESTAMP)]
    batch = DBatch(schema, cols, n)
    exprs = [
        Col("i") + Col("j") * 3 - 7, Col("i") // Col("j"), Col("i") % Col("j"), Col("i") / Col("j"),
        Col("f") // Col("g"), Col("f") % Col("g"), Col("f") * (Lit(1) - Col("g")) + Col("i"),
        (Col("i") > 10) & (Col("f") <= 2.5), (Col("i") == Col("j")) | (Col("f") != Col("g")),
        Col("s").like("%AIR%"), Col("s").like("_AI_"), Col("s").like("AIR"), Col("s") == "AIR", Col("s") >= "RAIL",
        Col("t") <= "1980-01-01",
    ]
    out = dev.eval_numeric(batch, exprs)
    py_rows = list(zip(i.tolist(), j.tolist(), [float(x) for x in f], [float(x) for x in g], words,
                       [datetime.utcfromtimestamp(x / 1e6) for x in ts.tolist()]))
    for expr, (col, tag) in zip(exprs, out):
        expr.infer_type(schema)  # the reference rewrites ISO literals next to timestamps here
        fn = compile_expr(expr, schema)
        want = [fn(row) for row in py_rows]
        got = col.data[:n].cpu().numpy().tolist()
        if tag == "B":
            assert [bool(v) for v in got] == [bool(v) for v in want], str(expr)
        else:
            assert got == want, str(expr)
    assert dev.read_flags() == 0
    dev.eval_numeric(batch, [Col("i") / (Col("j") - Col("j"))])
    assert dev.read_flags() & 1, "division by zero must raise the flag"
    dev.reset_flags()


def test_quantise_flags_overflow(dev):
    import torch

    from minispark_amd import hipspark as hs
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.device import DCol

    ok = dev.quantise_col(DCol(hs.F64, dev.to_device(np.array([0.1, -3.25e38, 1e-50]), torch.float64), 3), T.FLOAT)
    assert np.array_equal(ok.data.cpu().numpy(), np.array([0.1, -3.25e38, 1e-50], dtype=np.float32))
    assert dev.read_flags() == 0
    dev.quantise_col(DCol(hs.F64, dev.to_device(np.array([1e39]), torch.float64), 1), T.FLOAT)
    assert dev.read_flags() & hs.FLAG_FLT_OVERFLOW
    dev.reset_flags()
    dev.quantise_col(DCol(hs.I64, dev.to_device(np.array([2**31], dtype=np.int64), torch.int64), 1), T.INTEGER)
    assert dev.read_flags() & hs.FLAG_INT_OVERFLOW
    dev.reset_flags()
