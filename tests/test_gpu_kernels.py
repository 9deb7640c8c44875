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


@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 2047, 2048, 2049, 4095, 4096, 4097, 8193, 16383, 16384, 16385, 100_003, 3_000_017])
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


@pytest.mark.parametrize("n,p", [(0, 0.5), (1, 1.0), (17, 0.5), (4096, 0.5), (4097, 0.0), (4097, 1.0), (16384, 0.5), (16385, 1.0),
                                 (250_000, 0.3), (1_000_003, 0.9), (9_000_001, 0.5), (40_000_000, 0.02)])
def test_compact_is_stable_selection(dev, n, p):
    """Stable compaction of a byte mask (any non-zero byte keeps the row) against numpy.nonzero, up to thousands of tiles."""
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


@pytest.mark.parametrize("n_parts", [7, 10])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_partition_of_float_keys_follows_pythons_float_hash(dev, n_parts, dtype):
    """Round 3: FLOAT keys go to partition hash(float) % P with CPython's own (unrandomised) float hash - the partition decides
    which JoinJob sums a row (reference tasks.py:362), so it has to be the reference's.  Round 2 used it for integral values
    only and an arbitrary hash otherwise."""
    import torch

    from minispark_amd import hipspark as hs
    from minispark_amd.device import DCol

    rng = _rng(11)
    vals = np.concatenate([rng.normal(0, 1e3, 20_000), rng.normal(0, 1e-6, 2_000), rng.integers(-10**6, 10**6, 2_000).astype(np.float64),
                           rng.normal(0, 1, 2_000) * np.exp2(rng.integers(-200, 200, 2_000)),
                           [0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 1e300, -1e300, 5e-324, np.inf, -np.inf, 2.0**61, 2.0**61 - 1, -(2.0**62)]])
    keys = vals.astype(dtype)
    kind = hs.F32 if dtype == np.float32 else hs.F64
    col = DCol(kind, torch.from_numpy(keys).cuda(), len(keys))
    part = torch.empty(len(keys), dtype=torch.uint8, device="cuda")
    k = col.as_hs()
    import ctypes as C

    hs.check(dev.lib.hs_partition_ids(dev.stream, C.byref(k), None, len(keys), n_parts, part.data_ptr()), "hs_partition_ids")
    got = part.cpu().numpy()
    want = np.array([hash(float(x)) % n_parts for x in keys], dtype=np.uint8)
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]


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


@pytest.mark.parametrize("n_left,n_right,spread,seed", [(1, 5, 1, 0), (3000, 5000, 2, 1), (70_000, 200_000, 4, 2),
                                                        (2_000_000, 3_000_000, 10, 3), (300_000, 100_000, 30, 4)])
def test_dense_integer_join_matches_reference_order(dev, n_left, n_right, spread, seed):
    """Round 4: INTEGER keys over a dense range take hs_join_dense_* - two stable range-partition passes over the build
    rows, a CSR over key SLOTS assembled per partition in LDS (rows of a slot ascending, no sort, no global atomic), probe by
    two adjacent offsets.  Duplicates on both sides, keys without a partner on both sides, a negative key_min, zero / one /
    two partition passes: pairs must come out ordered by right row, then left row ascending (reference tasks.py:224-240)."""
    from minispark_amd.constants import ColumnType

    rng = _rng(100 + seed)
    span = max(1, n_left * spread // 2)
    base = -span // 3
    left = (base + rng.integers(0, span, n_left)).astype(np.int32)       # duplicates happen (birthday), gaps too
    if n_left > 1:
        left[rng.integers(0, n_left, max(1, n_left // 50))] = left[0]    # ... and one key many times
    right = (base - 5 + rng.integers(0, span + 10, n_right)).astype(np.int32)  # some probe keys fall outside the range
    lcol, rcol = dev.upload_raw(left, ColumnType.INTEGER), dev.upload_raw(right, ColumnType.INTEGER)
    before = getattr(dev, "dense_joins", 0)
    out_left, out_right, out_start, n_out = dev.join_indices(lcol, rcol)
    assert getattr(dev, "dense_joins", 0) == before + 1 and dev.last_join["mode"] == "dense csr"
    assert dev.read_flags() == 0
    order = np.argsort(left, kind="stable")
    keys_sorted = left[order]
    lo, hi = np.searchsorted(keys_sorted, right, "left"), np.searchsorted(keys_sorted, right, "right")
    counts = hi - lo
    want_right = np.repeat(np.arange(n_right, dtype=np.int64), counts)
    starts = np.concatenate([[0], np.cumsum(counts)])
    within = np.arange(int(counts.sum()), dtype=np.int64) - np.repeat(starts[:-1], counts)
    want_left = order[np.repeat(lo, counts) + within].astype(np.int64)
    assert n_out == len(want_right)
    assert np.array_equal(out_start[: n_right + 1].cpu().numpy(), starts)
    assert np.array_equal(out_right[:n_out].cpu().numpy(), want_right)
    assert np.array_equal(out_left[:n_out].cpu().numpy(), want_left)


@pytest.mark.parametrize("n_left,n_right,seed", [(1, 5, 0), (700, 5000, 1), (3000, 900, 2), (70_000, 200_000, 3),
                                                 (2_000_000, 3_000_000, 4), (9_000_000, 1_000_000, 5),
                                                 (20_000_000, 500_000, 6)])  # (past 19 M build rows: 1024-slot windows)
def test_sparse_integer_join_takes_the_hash_windows_and_matches_reference_order(dev, n_left, n_right, seed):
    """Round 4: INTEGER keys that are NOT dense (the whole int32 range, negative values) take hs_join_hash_* - the build
    rows moved into hash-window order by zero / one / two stable partition passes, every 512-slot window of the {key, word}
    table assembled in LDS, probe by one 8-byte slot read, second pass shared with the dense form.  Duplicates on both sides
    (one key many times), probe keys without a partner, INT32_MIN / INT32_MAX / 0 / -1 as keys: pairs must come out ordered by
    right row, then left row ascending (reference tasks.py:224-240)."""
    from minispark_amd.constants import ColumnType

    rng = _rng(200 + seed)
    pool = rng.integers(-2**31, 2**31 - 1, max(2, n_left * 3 // 4), dtype=np.int64).astype(np.int32)
    pool[:4] = np.array([-2**31, 2**31 - 1, 0, -1], dtype=np.int32)[: len(pool[:4])]
    left = pool[rng.integers(0, len(pool), n_left)]                       # duplicates happen, most pool keys appear
    if n_left > 1:
        left[rng.integers(0, n_left, max(1, n_left // 50))] = left[0]     # ... and one key many times
    right = pool[rng.integers(0, len(pool), n_right)]
    miss = rng.random(n_right) < 0.2
    right[miss] = rng.integers(-2**31, 2**31 - 1, int(miss.sum()), dtype=np.int64).astype(np.int32)  # (almost surely) no partner
    lcol, rcol = dev.upload_raw(left, ColumnType.INTEGER), dev.upload_raw(right, ColumnType.INTEGER)
    before = getattr(dev, "hashed_joins", 0)
    out_left, out_right, out_start, n_out = dev.join_indices(lcol, rcol)
    if n_left > 1:  # (a single build row spans one slot: the dense form holds it)
        assert getattr(dev, "hashed_joins", 0) == before + 1 and dev.last_join["mode"] == "hashed windows"
    assert dev.read_flags() == 0
    order = np.argsort(left, kind="stable")
    keys_sorted = left[order]
    lo, hi = np.searchsorted(keys_sorted, right, "left"), np.searchsorted(keys_sorted, right, "right")
    counts = hi - lo
    want_right = np.repeat(np.arange(n_right, dtype=np.int64), counts)
    starts = np.concatenate([[0], np.cumsum(counts)])
    within = np.arange(int(counts.sum()), dtype=np.int64) - np.repeat(starts[:-1], counts)
    want_left = order[np.repeat(lo, counts) + within].astype(np.int64)
    assert n_out == len(want_right)
    assert np.array_equal(out_start[: n_right + 1].cpu().numpy(), starts)
    assert np.array_equal(out_right[:n_out].cpu().numpy(), want_right)
    assert np.array_equal(out_left[:n_out].cpu().numpy(), want_left)


def test_a_hash_window_that_overflows_hands_the_join_to_the_global_table(dev):
    """More distinct keys in ONE 512-slot window than it has slots (keys picked by brute force to share a window - a
    degenerate hash, not a data pattern): the build leaves that window empty and says so, the engine's join falls back to
    hs_join_build and the pairs are still the reference's."""
    from minispark_amd.constants import ColumnType

    def mix32(k):
        k = k.astype(np.uint64)
        h = (k * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
        h ^= h >> np.uint64(15)
        h = (h * np.uint64(0x85EBCA77)) & np.uint64(0xFFFFFFFF)
        return h ^ (h >> np.uint64(16))

    n_left = 4000
    windows = (n_left * 7 // 4 + 511) >> 9  # (tables of up to 19 M build rows are cut into 512-slot windows)
    cand = np.arange(0, 4_000_000, dtype=np.int64)
    in_window_0 = cand[((mix32(cand) * np.uint64(windows)) >> np.uint64(32)) == 0]
    assert len(in_window_0) >= 1100
    left = np.concatenate([in_window_0[:1100], cand[:n_left - 1100] + 5_000_000]).astype(np.int32)
    # spread the keys far apart so that the dense form refuses them (a range of more than 32 slots per build row)
    left[-1] = 2_000_000_000
    right = np.concatenate([left[::3], np.array([7, -9], dtype=np.int32)])
    lcol, rcol = dev.upload_raw(left, ColumnType.INTEGER), dev.upload_raw(right, ColumnType.INTEGER)
    before = getattr(dev, "hashed_joins", 0)
    out_left, out_right, _, n_out = dev.join_indices(lcol, rcol)
    assert getattr(dev, "hashed_joins", 0) == before and dev.last_join["mode"] == "global hash table"
    assert dev.read_flags() == 0
    pos = {int(k): i for i, k in enumerate(left.tolist())}
    want = [(pos[int(k)], ri) for ri, k in enumerate(right.tolist()) if int(k) in pos]
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


@pytest.mark.parametrize("n,jit", [(4000, True), (4000, False), (4003, True), (1, True), (5, True)])
def test_expression_evaluator_against_python_semantics(dev, n, jit):
    """hs_eval vs the oracle's row evaluator (Python operators): mixed int/float arithmetic, floor division
    and modulo with negative operands, comparisons, & / |, LIKE, string comparison - through the compiled
    four-rows-per-lane kernels (row counts that are not multiples of 4 exercise the guarded tail) and through the
    interpreter kernel."""
    from datetime import datetime

    from minispark_amd.constants import ColumnType as T
    from minispark_amd.device import DBatch
    from minispark_amd.io import StrCol
    from minispark_amd.sql import Col, Lit
    from oracle.py_engine import compile_expr

    stats = (C.c_int32 * 3)()
    dev.lib.hs_jit_stats(stats)
    launched_before = stats[1]
    dev.lib.hs_jit_set_enabled(1 if jit else 0)
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
    dev.lib.hs_jit_stats(stats)
    dev.lib.hs_jit_set_enabled(1)
    assert (stats[1] > launched_before) == jit, "the compiled / interpreted route was not the one exercised"


@pytest.mark.parametrize("jit", [True, False])
def test_integer_floor_division_and_modulo_on_both_widths(dev, jit):
    """Integers are 64-bit in flight; the compiler narrows // and % of sign-extended 32-bit operands to a 32-bit division
    and keeps the wide one otherwise (products beyond 32 bits) - both must give Python's floor semantics, divisors of
    either sign, and INT32_MIN // -1 = 2147483648 (hs_floordiv_i / hs_mod_i in csrc/hs_device.h answer b == -1 without
    dividing: the narrowed division would return INT32_MIN)."""
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.device import DBatch
    from minispark_amd.sql import Col, Lit

    dev.lib.hs_jit_set_enabled(1 if jit else 0)
    r = _rng(23)
    n = 3001
    i = r.integers(-2**31, 2**31, n).astype(np.int32)
    i[:6] = [-2**31, 2**31 - 1, 0, -1, 1, -2**31 + 1]
    j = np.where(r.random(n) < 0.5, r.integers(1, 50_000, n), -r.integers(1, 50_000, n)).astype(np.int32)
    j[:6] = [-1, -1, 7, -1, -7, 2**31 - 1]
    t = (r.integers(-10**6, 10**6, n).astype(np.int64) * 86_400_000_123)
    schema = [("i", T.INTEGER), ("j", T.INTEGER), ("t", T.TIMESTAMP)]
    batch = DBatch(schema, [dev.upload_raw(i, T.INTEGER), dev.upload_raw(j, T.INTEGER), dev.upload_raw(t, T.TIMESTAMP)], n)
    exprs = [(Col("i") // Col("j"), lambda a, b, c: a // b), (Col("i") % Col("j"), lambda a, b, c: a % b),
             (Col("i") // Lit(-1), lambda a, b, c: a // -1), (Col("i") % Lit(-1), lambda a, b, c: a % -1),
             (Col("i") % Lit(97), lambda a, b, c: a % 97), (Col("i") // Lit(-97), lambda a, b, c: a // -97),
             (Col("i") * Col("j") // Col("j"), lambda a, b, c: a * b // b), ((Col("i") * 100_000 + Col("j")) % Col("j"), lambda a, b, c: (a * 100_000 + b) % b)]
    out = dev.eval_numeric(batch, [e for e, _ in exprs])
    rows = list(zip(i.tolist(), j.tolist(), t.tolist()))
    for (expr, fn), (col, _) in zip(exprs, out):
        assert col.data[:n].cpu().numpy().tolist() == [fn(*row) for row in rows], str(expr)
    assert dev.read_flags() == 0
    dev.lib.hs_jit_set_enabled(1)


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


@pytest.mark.parametrize("world,units,cap,groups", [(1, 5, 4, 3), (4, 9, 8, 6), (8, 36, 4, 3), (8, 20, 8, 6), (6, 12, 16, 20),
                                                      (2, 60, 16, 10)])
def test_finish_launch_merges_slabs_of_many_ranks_in_block_order(dev, world, units, cap, groups):
    """hs_agg_finish over hand-built slabs of `world` ranks (the 8-GPU shape cannot be launched here): every
    group's partials must be folded in ascending block id, in fp64 from f32 partials - compared bit for bit with
    a sequential Python fold; block b lives on rank b % world, units hold each key at most once, unused rows
    carry order key -1.  The last shape does not fit the all-in-LDS variant and runs from global memory."""
    import ctypes as C

    import torch

    from minispark_amd import hipspark as hs
    from minispark_amd.distributed import SlabLayout

    rng = _rng(world * 1000 + units)
    m = units * cap                                   # rows per slab
    layout = SlabLayout.build(m, [(4, torch.int32), (4, torch.float32), (4, torch.int32), (4, torch.float32)])
    slabs = np.zeros((world, layout.nbytes), dtype=np.uint8)
    keys_all = rng.choice(np.arange(-50, 50), size=groups, replace=False).astype(np.int32)
    partials: dict[int, list[tuple[int, float, int, float]]] = {int(k): [] for k in keys_all}
    n_blocks = world * units
    for b in range(n_blocks):
        rank, u = b % world, b // world
        present = keys_all[rng.random(groups) < 0.7][:cap]
        rng.shuffle(present)
        view = slabs[rank]
        order = view[layout.order_offset: layout.order_offset + 8 * m].view(np.int64)
        cols = [view[c.offset: c.offset + 4 * m] for c in layout.columns]
        order[u * cap: (u + 1) * cap] = -1
        for j, k in enumerate(present):
            row = u * cap + j
            f, i, mx = np.float32(rng.normal(0, 1e3)), int(rng.integers(-10**6, 10**6)), np.float32(rng.normal(0, 50))
            order[row] = b
            cols[0].view(np.int32)[row] = k
            cols[1].view(np.float32)[row] = f
            cols[2].view(np.int32)[row] = i
            cols[3].view(np.float32)[row] = mx
            partials[int(k)].append((b, float(f), i, float(mx)))
    slabs[:, 0:4].view(np.uint32)[:] = 0
    slabs[world - 1, 0:4].view(np.uint32)[0] = hs.FLAG_STR_TOO_LONG  # a remote status bit must reach the header

    desc = hs.hs_slab_desc()
    desc.slab_rows, desc.stride, desc.order_off, desc.key_off = m, layout.nbytes, layout.order_offset, layout.columns[0].offset
    desc.key_kind, desc.key_len, desc.n_acc = hs.I32, 0, 3
    for a, kind in enumerate([hs.F32, hs.I32, hs.F32]):
        desc.acc_off[a], desc.acc_kind[a] = layout.columns[1 + a].offset, kind
    fin = hs.hs_finish_spec()
    fin.n_fold = 3
    for j, (src, op) in enumerate([(0, hs.AGG_SUM), (1, hs.AGG_SUM), (2, hs.AGG_MAX)]):
        fin.fold_src[j], fin.fold_op[j] = src, op
    merge_cap = 4
    while merge_cap < 2 * groups:
        merge_cap *= 2
    outs = [(0, 0, hs.I32, 4), (1, 0, hs.F32, 4), (1, 1, hs.I64, 8), (1, 2, hs.F32, 4)]
    fin.n_out = len(outs)
    pos, offsets = 16, []
    for o, (src, index, kind, width) in enumerate(outs):
        fin.outs[o].src, fin.outs[o].index, fin.outs[o].kind, fin.outs[o].offset = src, index, kind, pos
        offsets.append(pos)
        pos = (pos + merge_cap * width + 15) & ~15
    gathered = dev.to_device(slabs.reshape(-1), torch.uint8)
    result = torch.zeros(pos + 64, dtype=torch.uint8, device=dev.device)
    scratch = dev.workspace(dev.lib.hs_agg_finish_scratch_bytes(merge_cap, 3))
    dev.flags.zero_()
    hs.check(dev.lib.hs_agg_finish(dev.stream, gathered.data_ptr(), world, C.byref(desc), C.byref(fin), None, n_blocks,
                                   merge_cap, result.data_ptr(), scratch.data_ptr(), dev.flags.data_ptr(), None),
             "hs_agg_finish")
    host = result.cpu().numpy()
    assert int(host[0:4].view(np.uint32)[0]) == hs.FLAG_STR_TOO_LONG and int(host[4:8].view(np.uint32)[0]) == 1
    assert int(dev.flags[0].item()) == 0  # handed over and reset
    ng = int(host[8:16].view(np.int64)[0])
    want = {k: v for k, v in partials.items() if v}
    assert ng == len(want)
    got_keys = host[offsets[0]: offsets[0] + 4 * ng].view(np.int32)
    got_f = host[offsets[1]: offsets[1] + 4 * ng].view(np.float32)
    got_i = host[offsets[2]: offsets[2] + 8 * ng].view(np.int64)
    got_mx = host[offsets[3]: offsets[3] + 4 * ng].view(np.float32)
    assert sorted(got_keys.tolist()) == sorted(want)
    for g in range(ng):
        rows = sorted(want[int(got_keys[g])])  # ascending block id
        acc_f, acc_i, acc_mx = 0.0, 0, float(-2**31)
        for _, f, i, mx in rows:
            acc_f += f
            acc_i += i
            acc_mx = mx if mx > acc_mx else acc_mx
        assert np.float32(acc_f) == got_f[g] and acc_i == int(got_i[g]) and np.float32(acc_mx) == got_mx[g]


@pytest.mark.parametrize("n,groups", [(5000, 5000), (200_000, 90), (200_000, 7), (300_000, 3000)])
def test_group_build_lists_are_ascending_whatever_their_length(dev, n, groups):
    """hs_group_build (the join's build): the positions of every key must come out ascending - one lane sorts short
    lists, a workgroup sorts long ones in LDS (<= 16 Ki rows) or in global memory (beyond)."""
    import torch

    from minispark_amd import hipspark as hs
    from minispark_amd.device import DCol

    keys = _rng(n + groups).integers(0, groups, n).astype(np.int32) * 7 - 3
    col = DCol(hs.I32, dev.to_device(keys, torch.int32), n)
    slot_start, positions, slot_list, ngr = dev._group_build(col, None, n)
    assert ngr == len(np.unique(keys))
    starts, pos, slots = slot_start.cpu().numpy(), positions.cpu().numpy(), slot_list[:ngr].cpu().numpy()
    seen = 0
    for s in slots:
        rows = pos[starts[s]: starts[s + 1]]
        assert len(rows) > 0 and np.all(np.diff(rows) > 0), "row list not strictly ascending"
        assert len(np.unique(keys[rows])) == 1
        seen += len(rows)
    assert seen == n
    assert dev.read_flags() == 0


@pytest.mark.parametrize("cap,holes", [(16, False), (16, True), (64, True), (256, False)])
def test_ordered_merge_survives_invalid_rows_and_full_dictionaries(dev, cap, holes):
    """hs_agg_merge with order keys (block ids): rows that are padding (order -1) or whose key did not fit a full
    dictionary leave holes inside a block's row range.  The visiting sequence must still be a permutation of the
    valid rows (it once assumed position = row - first row of the block: with holes the representatives of some
    groups were never written and stale indices reached the gathers).  With a dictionary large enough the result
    equals a sequential fold in (block, row) order; with a small one only HS_FLAG_MERGE_FULL may be raised."""
    import torch

    from minispark_amd import hipspark as hs
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.device import DBatch, DCol
    from minispark_amd.sql import AggCol, Col

    rng = _rng(cap + holes)
    blocks, per_block, groups = 7, 24, 50
    n = blocks * per_block
    keys = rng.integers(0, groups, n).astype(np.int32)
    for b in range(blocks):  # a key occurs at most once per block (like partial rows of a unit)
        keys[b * per_block: (b + 1) * per_block] = rng.permutation(groups)[:per_block]
    vals = rng.normal(0, 100, n).astype(np.float32)
    order = np.repeat(rng.permutation(blocks), per_block).astype(np.int64)  # blocks not in id order
    if holes:
        order[rng.random(n) < 0.3] = -1
    batch = DBatch([("k", T.INTEGER), ("v", T.FLOAT)], [DCol(hs.I32, dev.to_device(keys, torch.int32), n),
                                                        DCol(hs.F32, dev.to_device(vals, torch.float32), n)], n,
                   order=dev.to_device(order, torch.int64), total_units=blocks)
    dev.reset_flags()
    out = dev.aggregate_merge(batch, [AggCol("sum", Col("v"))], [("k", T.INTEGER), ("v", T.FLOAT)], cap)
    torch.cuda.synchronize()
    flags = dev.read_flags()
    valid = order >= 0
    n_groups = len(np.unique(keys[valid]))
    assert not flags & hs.FLAG_BAD_PROGRAM, "a group was left without its representative row"
    ng = int(out.nrows_dev[0].item())
    got_keys = out.cols[0].data[:ng].cpu().numpy()
    if n_groups > cap:
        assert flags == hs.FLAG_MERGE_FULL and ng <= cap and set(got_keys.tolist()) <= set(keys[valid].tolist())
        dev.reset_flags()
        return
    assert flags == 0 and ng == n_groups
    got = dict(zip(got_keys.tolist(), out.cols[1].data[:ng].cpu().numpy().tolist()))
    for key in np.unique(keys[valid]):
        rows = [r for r in np.argsort(order, kind="stable") if valid[r] and keys[r] == key]  # (block id, row) order
        acc = 0.0
        for r in rows:
            acc += float(vals[r])
        assert got[int(key)] == acc, key


def test_copy_segments_packs_and_unpacks_byte_ranges(dev):
    """hs_copy_segments (the exchange's pack / unpack launch): aligned, misaligned, tiny, empty and large segments."""
    import torch

    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, 3_000_000, dtype=np.uint8)
    d_src = dev.to_device(src)
    d_dst = torch.zeros(3_100_000, dtype=torch.uint8, device=dev.device)
    #        (source offset, destination offset, bytes)
    plan = [(0, 0, 1_000_000), (1_000_000, 1_000_016, 999_983), (1_999_983, 2_000_003, 17), (2_000_000, 2_000_100, 0),
            (2_000_000, 2_000_033, 5), (2_100_000, 2_100_007, 700_001), (2_999_999, 3_099_999, 1)]
    dev.copy_segments([(d_src.data_ptr() + s, d_dst.data_ptr() + d, n) for s, d, n in plan])
    want = np.zeros(3_100_000, dtype=np.uint8)
    for s, d, n in plan:
        want[d: d + n] = src[s: s + n]
    assert np.array_equal(d_dst.cpu().numpy(), want)
    dev.copy_segments([])  # nothing to do


@pytest.mark.parametrize("n,n_parts,skew", [(1, 1, False), (2047, 10, False), (2049, 10, True), (70_001, 32, False),
                                            (1_000_003, 10, True), (300_000, 3, False)])
def test_partition_perm_is_the_stable_counting_sort(n, n_parts, skew):
    """hs_partition_perm (reference tasks.py:347-375: rows go to their shuffle partition in row order) against numpy's stable
    argsort: the permutation and the partition starts; round 3 rewrote the scatter (striped rows, ballot ranks, LDS staging)."""
    import torch

    from minispark_amd.device import Device

    dev = Device(0)
    rng = np.random.default_rng(n + n_parts)
    ids = (rng.integers(0, n_parts, n) if not skew else np.minimum(rng.geometric(0.4, n) - 1, n_parts - 1)).astype(np.uint8)
    perm, start = dev.partition_by_ids(torch.from_numpy(ids).cuda(), n, n_parts)
    torch.cuda.synchronize()
    want = np.argsort(ids, kind="stable")
    assert np.array_equal(perm[:n].cpu().numpy(), want)
    assert start == [int(v) for v in np.concatenate([[0], np.cumsum(np.bincount(ids, minlength=n_parts))])]
