"""The HBM tier of GROUP BY by radix partitioning (csrc/hs_radix.hip: hs_group_radix_plan / run / emit) against a
numpy restatement of the reference's per-block dictionary fold (tasks.py:284-310): every group's values are folded in
ascending row order in fp64 / int, so ``np.add.at`` (unbuffered, in index order) IS the expected result - bit for bit,
also through the f32 / i32 quantisation of the shuffle write."""

from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from minispark_amd.execution import HipExecutionEngine

    with HipExecutionEngine(0) as e:
        yield e.dev


def _expected(keys, vals, ops, is_int, bounds, quantise):
    """-> per unit (sorted distinct keys, one accumulator array per aggregate in that order)"""
    from minispark_amd import hipspark as hs

    out = []
    for lo, hi in zip(bounds, bounds[1:]):
        uniq, inv = np.unique(keys[lo:hi], return_inverse=True)
        accs = []
        for v, op, integer in zip(vals, ops, is_int):
            x = (np.full(hi - lo, v) if np.isscalar(v) else v[lo:hi]).astype(np.int64 if integer else np.float64)
            if op == hs.AGG_SUM:
                acc = np.zeros(len(uniq), x.dtype)
                np.add.at(acc, inv, x)
            elif op == hs.AGG_MIN:
                acc = np.full(len(uniq), 2147483647, x.dtype)
                np.minimum.at(acc, inv, x)
            else:
                acc = np.full(len(uniq), -2147483648, x.dtype)
                np.maximum.at(acc, inv, x)
            if quantise:
                acc = acc.astype(np.int32 if integer else np.float32)
            accs.append(acc)
        out.append((uniq.astype(np.int64), accs))
    return out


def _run(dev, keys, key_kind, sel, bounds, values, ops, quantise):
    import torch

    from minispark_amd import hipspark as hs
    from minispark_amd.device import DCol

    tk = torch.from_numpy(keys).cuda()
    key = DCol(key_kind, tk, len(keys))
    tsel = torch.from_numpy(sel).cuda() if sel is not None else None
    n = len(sel) if sel is not None else len(keys)
    vals = []
    for v in values:
        if np.isscalar(v):
            vals.append((None, int(v) & (2**64 - 1), True))
        else:
            kind = {np.dtype(np.int32): hs.I32, np.dtype(np.float32): hs.F32, np.dtype(np.int64): hs.I64,
                    np.dtype(np.float64): hs.F64}[v.dtype]
            vals.append((DCol(kind, torch.from_numpy(v).cuda(), len(v)), 0, kind in (hs.I32, hs.I64)))
    tb = torch.from_numpy(np.asarray(bounds, dtype=np.int64)).cuda()
    biggest = max(b - a for a, b in zip(bounds, bounds[1:]))
    done = dev.group_radix(key, tsel, n, tb, len(bounds) - 1, biggest, vals, ops, quantise)
    assert done is not None, "a partition outgrew its dictionary"
    key_col, accs, unit_rows = done
    torch.cuda.synchronize()
    assert dev.read_flags() == 0
    k = key_col.data[: key_col.n].cpu().numpy().astype(np.int64)
    a = [c.data[: c.n].cpu().numpy() for c in accs]
    got = []
    for lo, hi in zip(unit_rows, unit_rows[1:]):
        order = np.argsort(k[lo:hi], kind="stable")
        got.append((k[lo:hi][order], [x[lo:hi][order] for x in a]))
    return got


def _bits(x):
    return x.view({4: np.int32, 8: np.int64}[x.dtype.itemsize])


def _same(got, want):
    """Bit for bit (signed zeros included), group for group - as arrays: these tests hold millions of groups."""
    assert len(got) == len(want)
    for u, ((gk, ga), (wk, wa)) in enumerate(zip(got, want)):
        assert len(gk) == len(wk), f"unit {u}: {len(gk)} groups, expected {len(wk)}"
        assert len(gk) < 2 or (gk[1:] != gk[:-1]).all(), f"unit {u}: a key twice in one unit"
        assert np.array_equal(gk, wk), f"unit {u}: other keys"
        for j, (x, y) in enumerate(zip(ga, wa)):
            if x.dtype != y.dtype and x.dtype.kind in "iu" and y.dtype.kind in "iu":
                x, y = x.astype(np.int64), y.astype(np.int64)
            assert x.dtype == y.dtype, (u, j, x.dtype, y.dtype)
            bad = np.nonzero(_bits(x) != _bits(y))[0]
            assert len(bad) == 0, (u, j, int(gk[bad[0]]), x[bad[0]].item(), y[bad[0]].item())


@pytest.mark.parametrize(("n", "units", "groups", "seed"), [
    (1, 1, 1, 0), (700, 3, 40, 1), (50_000, 4, 30_000, 2), (300_000, 3, 200_000, 3),  # one pass
    (3_000_000, 1, 700_000, 4), (3_000_000, 2, 50, 5), (2_500_000, 5, 2_000_000, 6),   # two passes; few / many groups
])
@pytest.mark.parametrize("quantise", [True, False])
def test_radix_tier_matches_the_ordered_fold(dev, n, units, groups, seed, quantise):
    from minispark_amd import hipspark as hs

    rng = np.random.default_rng(seed)
    keys = rng.integers(-groups // 2, groups - groups // 2, n).astype(np.int32)
    f = (rng.normal(0, 1, n) * np.exp2(rng.integers(-8, 8, n))).astype(np.float32)
    d = rng.normal(0, 1e3, n)  # an evaluated expression: fp64 cells
    i = rng.integers(-1000, 1000, n).astype(np.int32)
    cuts = np.sort(rng.integers(0, n + 1, units - 1)).tolist()
    bounds = [0] + cuts + [n]  # units of any size, empty ones included
    values = [f, d, i, 1, f, i]
    ops = [hs.AGG_SUM, hs.AGG_SUM, hs.AGG_SUM, hs.AGG_SUM, hs.AGG_MIN, hs.AGG_MAX]
    is_int = [False, False, True, True, False, True]
    got = _run(dev, keys, hs.I32, None, bounds, values, ops, quantise)
    _same(got, _expected(keys, values, ops, is_int, bounds, quantise))


@pytest.mark.parametrize("classes", ["f", "i", "c", "fc", "if", "cf", "ffc", "icf", "cii", "fif"])
@pytest.mark.parametrize(("n", "units", "groups"), [(60_000, 4, 9_000), (3_000_000, 2, 400_000), (400_000, 3, 41)])
def test_radix_sum_fold_in_every_class_combination(dev, classes, n, units, groups):
    """Round 3: SUMs over f32 / i32 columns and integer constants (what SUM, AVG and COUNT lower to) run a fold specialised
    at compile time per class combination (k_rx_fold_sum), INTEGER keys the 4-byte partition kernels - same order of
    additions, so still bit for bit the reference's fold; one and two partition passes, ragged units; 41 groups: steps in
    which dozens of the 64 rows share a group (the fold's register-chain form)."""
    from minispark_amd import hipspark as hs

    rng = np.random.default_rng(len(classes) * 7 + n % 13)
    keys = rng.integers(-groups // 2, groups - groups // 2, n).astype(np.int32)
    cuts = np.sort(rng.integers(0, n + 1, units - 1)).tolist()
    bounds = [0] + cuts + [n]
    values, is_int = [], []
    for c in classes:
        if c == "f":
            values.append((rng.normal(0, 1, n) * np.exp2(rng.integers(-10, 10, n))).astype(np.float32))
        elif c == "i":
            values.append(rng.integers(-100_000, 100_000, n).astype(np.int32))
        else:
            values.append(1 if len(values) % 2 == 0 else 3)
        is_int.append(c != "f")
    ops = [hs.AGG_SUM] * len(values)
    for quantise in (True, False):
        got = _run(dev, keys, hs.I32, None, bounds, values, ops, quantise)
        _same(got, _expected(keys, values, ops, is_int, bounds, quantise))


def test_radix_tier_takes_float_and_short_string_keys(dev):
    """Round 3: keys whose 64-bit key word is the key itself - stored FLOAT, computed f64 (0.0 and -0.0 one group, as in
    a Python dict), STRING columns of one fixed length <= 7 bytes; and strings of 8 .. 16 bytes as TWO key words compared exactly (no hashing) -
    take the radix tier too; longer or variable-length strings are refused by the plan (the engine keeps the hash-table
    tier for those)."""
    import ctypes as C

    import torch

    from minispark_amd import hipspark as hs
    from minispark_amd.device import DCol

    rng = np.random.default_rng(77)
    n, bounds = 500_000, [0, 170_000, 170_000, 500_000]
    v = rng.normal(0, 10, n).astype(np.float32)
    values, ops, is_int = [v, 1], [hs.AGG_SUM, hs.AGG_SUM], [False, True]
    pool = np.concatenate([rng.normal(0, 100, 40_000), [0.0, -0.0, 1e300, -1e-300]])
    for kind, dtype in ((hs.F64, np.float64), (hs.F32, np.float32)):
        keys = pool[rng.integers(0, len(pool), n)].astype(dtype)
        tb = torch.from_numpy(np.asarray(bounds, dtype=np.int64)).cuda()
        done = dev.group_radix(DCol(kind, torch.from_numpy(keys).cuda(), n), None, n, tb, len(bounds) - 1, 330_000,
                               [(DCol(hs.F32, torch.from_numpy(v).cuda(), n), 0, False), (None, 1, True)], ops, True)
        assert done is not None
        key_col, accs, unit_rows = done
        k = key_col.data[: key_col.n].cpu().numpy()
        a = [c.data[: c.n].cpu().numpy() for c in accs]
        assert k.dtype == dtype
        for u, (lo, hi) in enumerate(zip(unit_rows, unit_rows[1:])):
            got = {float(k[i]): (a[0][i].item(), a[1][i].item()) for i in range(lo, hi)}
            assert len(got) == hi - lo
            want: dict = {}
            for key, x in zip(keys[bounds[u]:bounds[u + 1]].tolist(), v[bounds[u]:bounds[u + 1]].astype(np.float64).tolist()):
                s, c = want.get(key, (0.0, 0))  # a Python dict: 0.0 and -0.0 meet in one entry, sums in row order
                want[key] = (s + x, c + 1)
            assert got.keys() == want.keys()
            for key, (s, c) in want.items():
                assert got[key] == (np.float32(s).item(), c), (kind, u, key)
    # strings of one fixed length
    for length in (1, 3, 7, 8, 12, 16):  # <= 7: one packed key word; 8 .. 16: two key words per tuple, compared on both
        alphabet = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghij0123456789", dtype=np.uint8)
        words = alphabet[rng.integers(0, len(alphabet), (min(30_000, len(alphabet) ** length), length))]
        if length > 8:
            words[: len(words) // 2, :8] = words[0, :8]  # many keys that agree in their first word
        rows = words[rng.integers(0, len(words), n)]
        data = torch.from_numpy(np.ascontiguousarray(rows).reshape(-1)).cuda()
        lens = torch.full((n,), length, dtype=torch.uint8, device="cuda")
        key = DCol(hs.STR, data, n, lens=lens, offs=None, fixed_len=length)
        tb = torch.from_numpy(np.asarray(bounds, dtype=np.int64)).cuda()
        done = dev.group_radix(key, None, n, tb, len(bounds) - 1, 330_000,
                               [(DCol(hs.F32, torch.from_numpy(v).cuda(), n), 0, False), (None, 1, True)], ops, True)
        assert done is not None
        key_col, accs, unit_rows = done
        assert key_col.kind == hs.STR and key_col.fixed_len == length
        kb = key_col.data[: key_col.n * length].cpu().numpy().reshape(-1, length)
        a = [c.data[: c.n].cpu().numpy() for c in accs]
        as_int = lambda m: [bytes(r) for r in m]  # noqa: E731
        for u, (lo, hi) in enumerate(zip(unit_rows, unit_rows[1:])):
            got = {key: (a[0][lo + i].item(), a[1][lo + i].item()) for i, key in enumerate(as_int(kb[lo:hi]))}
            assert len(got) == hi - lo
            want = {}
            for key, x in zip(as_int(rows[bounds[u]:bounds[u + 1]]), v[bounds[u]:bounds[u + 1]].astype(np.float64).tolist()):
                s, c = want.get(key, (0.0, 0))
                want[key] = (s + x, c + 1)
            assert got.keys() == want.keys()
            for key, (s, c) in want.items():
                assert got[key] == (np.float32(s).item(), c), (length, u, key)
    plan, spec = hs.hs_radix_plan(), hs.hs_agg_spec()
    spec.n_acc, spec.op[0], spec.is_int[0] = 1, hs.AGG_SUM, 0
    kinds = (C.c_int32 * 1)(hs.F32)
    assert dev.lib.hs_group_radix_plan(hs.STR + 256 * 17, 1000, 1, 1000, kinds, C.byref(spec), 1, C.byref(plan)) == 2  # HS_E_LIMIT
    assert dev.lib.hs_group_radix_plan(hs.STR + 256 * 12, 1000, 1, 1000, kinds, C.byref(spec), 1, C.byref(plan)) == 0
    spec.op[0] = hs.AGG_MIN  # two-word keys only know the SUM-specialised fold
    assert dev.lib.hs_group_radix_plan(hs.STR + 256 * 12, 1000, 1, 1000, kinds, C.byref(spec), 1, C.byref(plan)) == 2


def test_radix_tier_over_a_selection_and_timestamp_keys(dev):
    """WHERE survivors arrive as an ascending row list; the key is read through it, values are position-indexed."""
    from minispark_amd import hipspark as hs

    rng = np.random.default_rng(9)
    rows = 400_000
    all_keys = (rng.integers(0, 90_000, rows) * 86_400_000_000 - 2**40).astype(np.int64)
    sel = np.flatnonzero(rng.random(rows) < 0.4).astype(np.int64)
    n = len(sel)
    v = rng.normal(0, 50, n).astype(np.float32)
    unit_row_bounds = [0, 150_000, 150_000, 310_000, rows]
    bounds = [int(np.searchsorted(sel, b)) for b in unit_row_bounds]
    values, ops, is_int = [v, 1], [hs.AGG_SUM, hs.AGG_SUM], [False, True]
    got = _run(dev, all_keys, hs.I64, sel, bounds, values, ops, True)
    _same(got, _expected(all_keys[sel], values, ops, is_int, bounds, True))


def test_radix_tier_reports_an_outgrown_dictionary(dev):
    """Every row its own group and a unit far bigger than declared: partitions hold more keys than a dictionary has
    slots -> the wrapper answers None (the engine then takes the hash-table path)."""
    import torch

    from minispark_amd import hipspark as hs
    from minispark_amd.device import DCol

    n = 100_000
    key = DCol(hs.I32, torch.arange(n, dtype=torch.int32, device="cuda"), n)
    bounds = torch.tensor([0, n], dtype=torch.int64, device="cuda")
    assert dev.group_radix(key, None, n, bounds, 1, 2000, [(None, 1, True)], [hs.AGG_SUM], True) is None


def test_engine_takes_the_radix_tier_for_integer_keys(tmp_path):
    """Through the engine: the high-cardinality GROUP BY of test_gpu_q1_large runs on the radix tier and still equals
    the Python oracle bit for bit; with `Device.radix_enabled` off the hash-table tier gives the same rows."""
    from minispark_amd import constants
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.dataframe import DataFrame
    from minispark_amd.execution import HipExecutionEngine
    from minispark_amd.io import BlockFile
    from minispark_amd.sql import Col, Functions as F
    from oracle.py_engine import run_query
    from tests.conftest import assert_rows_match

    constants.SHUFFLE_FOLDER = tmp_path / "shuffle"
    rng = np.random.default_rng(5)
    n = 120_000
    keys = rng.integers(-20_000, 20_000, n).astype(np.int32)
    qty = rng.integers(1, 50, n).astype(np.int32)
    price = (rng.integers(100, 100_000, n) / 7.0).astype(np.float32)
    path = tmp_path / "t.bin"
    schema = [("k", T.INTEGER), ("q", T.INTEGER), ("p", T.FLOAT)]
    per = 50_000
    BlockFile(path).write_raw_blocks(schema, [[keys[i: i + per], qty[i: i + per], price[i: i + per]] for i in range(0, n, per)])

    def build(engine):
        return (DataFrame(engine).table(str(path)).filter(Col("q") > 3).group_by(Col("k"))
                .agg(F.sum(Col("q") * Col("p")).alias("rev"), F.count(), F.min(Col("p")).alias("lo"),
                     F.avg(Col("p")).alias("mean"), F.max(Col("q")).alias("hi")))

    want = run_query(build(object()).task)
    with HipExecutionEngine(0) as engine:
        rows = build(engine).collect()
        assert engine.dev.last_global_tier == "radix"
        assert_rows_match(rows, want, max_ulps=0)
        assert_rows_match(build(engine).collect(), want, max_ulps=0)
        engine.dev.radix_enabled = False
        engine._version += 1
        assert_rows_match(build(engine).collect(), want, max_ulps=0)
        assert engine.dev.last_global_tier == "hash"


@pytest.mark.parametrize(("n", "n_order"), [(1, 1), (5000, 3), (300_000, 287), (1_000_000, 70_000)])
def test_sort_by_order_is_a_stable_sort_with_padding_first(dev, n, n_order):
    """hs_sort_by_order (the merge order of a multi-rank final aggregate) against numpy's stable argsort."""
    import torch

    rng = np.random.default_rng(n)
    order = rng.integers(-1, n_order, n).astype(np.int64)
    sel, left = dev.sort_by_order(torch.from_numpy(order).cuda(), n, n_order)
    want = np.argsort(order, kind="stable")
    n_pad = int((order < 0).sum())
    assert left == n - n_pad
    assert np.array_equal(sel[:left].cpu().numpy(), want[n_pad:])


def test_unit_ids_per_row(dev):
    unit_rows = [0, 3, 3, 10, 11, 4000]
    ids = [7, 1, 300, 2, 9]
    got = dev.unit_ids_per_row(unit_rows, ids).cpu().numpy()
    want = np.repeat(np.asarray(ids), np.diff(unit_rows))
    assert np.array_equal(got, want)


def _hc_table(path, n, blocks, seed):
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.io import BlockFile

    rng = np.random.default_rng(seed)
    distinct = int(rng.choice([6_000, 20_000, n]))
    cols = {
        "k": rng.integers(-distinct // 2, distinct - distinct // 2, n).astype(np.int32),
        "t": (rng.integers(0, distinct, n) * 3_600_000_000 + 946_684_800_000_000).astype(np.int64),
        "f": rng.normal(0, 1e3, n).astype(np.float32),
        "g": rng.uniform(0, 1, n).astype(np.float32),
        "i": rng.integers(-10**5, 10**5, n).astype(np.int32),
    }
    schema = [("k", T.INTEGER), ("t", T.TIMESTAMP), ("f", T.FLOAT), ("g", T.FLOAT), ("i", T.INTEGER)]
    bounds = sorted({0, n, *rng.integers(1, n, blocks - 1).tolist()})
    BlockFile(path).write_raw_blocks(schema, [[c[lo:hi] for c in cols.values()] for lo, hi in zip(bounds, bounds[1:])])


def _hc_query(rng, api, path):
    C, F, Lit = api.Col, api.F, api.Lit
    df = api.DataFrame().table(path)
    if rng.random() < 0.6:
        df = df.filter(rng.choice([C("g") > 0.3, C("i") % 3 != 0, (C("f") < 500.0) & (C("g") <= 0.9)]))
    key = rng.choice(["k", "t", "m"])
    if key == "m":
        df = df.select((C("k") * 3 + C("i") % 2).alias("m"), C("f"), C("g"), C("i"))
    pool = [lambda: F.sum(C("f")), lambda: F.sum(C("i")), lambda: F.min(C("f")), lambda: F.max(C("i")), lambda: F.avg(C("g")),
            lambda: F.sum(C("f") * (Lit(1) - C("g"))), lambda: F.min(C("i")), lambda: F.max(C("g")), lambda: F.avg(C("i"))]
    aggs = [fn().alias(f"a{n}") for n, fn in enumerate(rng.sample(pool, rng.randint(1, 5)))]
    if rng.random() < 0.7:
        aggs.append(F.count())
    return df.group_by(C(key)).agg(*aggs)


_HC_FIRST = int(__import__("os").environ.get("HIPSPARK_HC_FIRST", "0"))


@pytest.mark.parametrize("seed", list(range(_HC_FIRST, _HC_FIRST + int(__import__("os").environ.get("HIPSPARK_HC_SEEDS", "8")))))
def test_random_high_cardinality_queries_match_the_oracle(tmp_path, seed):
    """Random GROUP BY queries on INTEGER / TIMESTAMP / computed integer keys with thousands of values per block:
    the engine ends up on the HBM tier, which for these keys is the radix partition - partial aggregate AND final merge -
    and must equal the Python oracle bit for bit (every group is folded in the reference's order)."""
    import random

    from minispark_amd import constants
    from minispark_amd.dataframe import DataFrame
    from minispark_amd.execution import HipExecutionEngine
    from minispark_amd.sql import Col, Functions, Lit
    from oracle.py_engine import run_query
    from tests.conftest import assert_rows_match
    from tests.queries import api_namespace

    constants.SHUFFLE_FOLDER = tmp_path / "shuffle"
    rng = random.Random(4100 + seed)
    path = tmp_path / "hc.bin"
    _hc_table(path, rng.choice([30_000, 60_000]), rng.choice([1, 3, 5]), seed)
    want = run_query(_hc_query(random.Random(seed), api_namespace(lambda: DataFrame(object()), Col, Functions, Lit), str(path)).task)
    with HipExecutionEngine(0) as engine:
        frame = _hc_query(random.Random(seed), api_namespace(lambda: DataFrame(engine), Col, Functions, Lit), str(path))
        for _ in range(2):
            assert_rows_match(frame.collect(), want, max_ulps=0)
        if seed < 8:  # the committed sample is sized for the HBM tier; a wider hunt may meet queries the LDS tiers hold
            assert engine._global_partial
        if engine._global_partial:
            assert engine.dev.last_global_tier == "radix"


@pytest.mark.parametrize("key,length", [("x", 0), ("q", 0), ("s", 5), ("s", 10), ("s", 16)])
@pytest.mark.parametrize("sums_only", [True, False])
def test_engine_takes_float_and_fixed_length_string_keys_to_the_hbm_tier(tmp_path, key, length, sums_only):
    """Round 3: GROUP BY a stored FLOAT column, a computed FLOAT expression or a STRING column whose rows all have one
    length (<= 16 bytes), thousands of values per block, through the whole engine: the radix tier where it holds the shape
    (wide strings: SUM / AVG / COUNT only - otherwise the hash-table tier), equal to the Python oracle bit for bit."""
    from minispark_amd import constants
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.dataframe import DataFrame
    from minispark_amd.execution import HipExecutionEngine
    from minispark_amd.io import BlockFile, StrCol
    from minispark_amd.sql import Col, Functions as F
    from oracle.py_engine import run_query
    from tests.conftest import assert_rows_match

    constants.SHUFFLE_FOLDER = tmp_path / "shuffle"
    rng = np.random.default_rng(31 + length)
    n, distinct = 50_000, 9_000
    pool = np.round(rng.normal(0, 100, distinct), 2).astype(np.float32)
    alphabet = "abcdefghijklmnopqrstuvwxyzABCDEF0123456789"
    words = ["".join(alphabet[c] for c in rng.integers(0, len(alphabet), max(length, 1))) for _ in range(distinct)]
    pick = rng.integers(0, distinct, n)
    cols = {"x": pool[pick], "s": [words[p] for p in pick], "f": rng.normal(0, 1e3, n).astype(np.float32),
            "i": rng.integers(-10**4, 10**4, n).astype(np.int32)}
    schema = [("x", T.FLOAT), ("s", T.STRING), ("f", T.FLOAT), ("i", T.INTEGER)]
    cuts = [0, 17_000, 17_001, 40_000, n]
    BlockFile(tmp_path / "t.bin").write_raw_blocks(schema, [[cols["x"][lo:hi], StrCol.from_strings(cols["s"][lo:hi]), cols["f"][lo:hi],
                                                            cols["i"][lo:hi]] for lo, hi in zip(cuts, cuts[1:])])

    def query(engine):
        df = DataFrame(engine).table(str(tmp_path / "t.bin")).filter(Col("i") % 7 != 0)
        if key == "q":  # a computed FLOAT key: an f64 that exists nowhere in the file
            df = df.select((Col("x") * 0.5 + 1.25).alias("q"), Col("f"), Col("i"))
        aggs = [F.sum(Col("f")).alias("sf"), F.avg(Col("i")).alias("ai"), F.count()]
        if not sums_only:
            aggs += [F.min(Col("f")).alias("lo"), F.max(Col("i")).alias("hi")]
        return df.group_by(Col(key)).agg(*aggs)

    want = run_query(query(object()).task)
    assert len(want) > 4096
    with HipExecutionEngine(0) as engine:
        frame = query(engine)
        for _ in range(2):
            assert_rows_match(frame.collect(), want, max_ulps=0)
        assert engine._global_partial
        if key != "s" or length <= 7 or sums_only:
            assert engine.dev.last_global_tier == "radix", engine.dev.last_global_tier


_HS_FIRST = int(__import__("os").environ.get("HIPSPARK_HS_FIRST", "0"))


@pytest.mark.parametrize("seed", list(range(_HS_FIRST, _HS_FIRST + int(__import__("os").environ.get("HIPSPARK_HS_SEEDS", "6")))))
def test_random_high_cardinality_queries_on_string_and_float_keys(tmp_path, seed):
    """The random high-cardinality queries again, grouped by a FLOAT column, a computed FLOAT or a STRING column of one fixed
    length (3 ... 16 bytes): whatever HBM tier the shape gets (radix where it holds it, the hash table otherwise), the rows
    must equal the Python oracle bit for bit."""
    import random

    from minispark_amd import constants
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.dataframe import DataFrame
    from minispark_amd.execution import HipExecutionEngine
    from minispark_amd.io import BlockFile, StrCol
    from minispark_amd.sql import Col, Functions as F, Lit
    from oracle.py_engine import run_query
    from tests.conftest import assert_rows_match

    constants.SHUFFLE_FOLDER = tmp_path / "shuffle"
    rng, nr = random.Random(9100 + seed), np.random.default_rng(9100 + seed)
    n = rng.choice([20_000, 45_000])
    distinct = rng.choice([5_000, 12_000, n])
    length = rng.choice([3, 6, 8, 11, 12, 16])
    alphabet = "abcdefghijklmnopqrstuvwxyz0123456789"
    words = ["".join(alphabet[c] for c in nr.integers(0, len(alphabet), length)) for _ in range(distinct)]
    pool = np.round(nr.normal(0, 1e3, distinct), 1).astype(np.float32)
    pick = nr.integers(0, distinct, n)
    cols = [pool[pick], StrCol.from_strings([words[p] for p in pick]), nr.normal(0, 1e3, n).astype(np.float32),
            nr.uniform(0, 1, n).astype(np.float32), nr.integers(-10**5, 10**5, n).astype(np.int32)]
    schema = [("x", T.FLOAT), ("s", T.STRING), ("f", T.FLOAT), ("g", T.FLOAT), ("i", T.INTEGER)]
    bounds = sorted({0, n, *nr.integers(1, n, rng.choice([1, 3, 5]) - 1).tolist()})

    def cut(c, lo, hi):
        return StrCol.from_strings(c.to_list()[lo:hi]) if isinstance(c, StrCol) else c[lo:hi]

    BlockFile(tmp_path / "t.bin").write_raw_blocks(schema, [[cut(c, lo, hi) for c in cols] for lo, hi in zip(bounds, bounds[1:])])

    def query(engine):
        qr = random.Random(seed)
        df = DataFrame(engine).table(str(tmp_path / "t.bin"))
        if qr.random() < 0.6:
            df = df.filter(qr.choice([Col("g") > 0.3, Col("i") % 3 != 0, (Col("f") < 500.0) & (Col("g") <= 0.9)]))
        key = qr.choice(["x", "s", "s", "q"])
        if key == "q":
            # a computed FLOAT key whose f64 values stay distinct after the f32 rounding of the shuffle write (powers of two
            # scale exactly): keys that collapse there are a documented divergence, not this test's subject
            df = df.select((Col("x") * qr.choice([0.5, 2.0, -4.0])).alias("q"), Col("f"), Col("g"), Col("i"))
        pool_ = [lambda: F.sum(Col("f")), lambda: F.sum(Col("i")), lambda: F.min(Col("f")), lambda: F.max(Col("i")),
                 lambda: F.avg(Col("g")), lambda: F.sum(Col("f") * (Lit(1) - Col("g"))), lambda: F.avg(Col("i"))]
        aggs = [fn().alias(f"a{k}") for k, fn in enumerate(qr.sample(pool_, qr.randint(1, 3)))]
        if qr.random() < 0.7:
            aggs.append(F.count())
        return df.group_by(Col(key)).agg(*aggs)

    want = run_query(query(object()).task)
    with HipExecutionEngine(0) as engine:
        frame = query(engine)
        for _ in range(2):
            assert_rows_match(frame.collect(), want, max_ulps=0)
        if engine._global_partial:
            assert engine.dev.last_global_tier in ("radix", "hash")
