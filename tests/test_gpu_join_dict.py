"""Round-2 operators: dictionary-coded string columns (encode / LIKE / compare / concat / GROUP BY / decode) and the
in-place unique-key join with computed units, each against numpy or the oracles; then BASELINE configs 4 and 5 at
sf=1 and sf=10 against the C ports of the reference's algorithm (oracle/q45_oracle.c, pinned to reference goldens)."""

from __future__ import annotations

import ctypes as C
import random

import numpy as np
import pytest

from tests.conftest import assert_rows_match

pytestmark = pytest.mark.gpu


@pytest.fixture()
def engine():
    from minispark_amd.execution import HipExecutionEngine

    with HipExecutionEngine(device=0) as e:
        yield e


def _api(engine):
    from minispark_amd.workloads import engine_api

    return engine_api(engine)


def _oracle_api():
    from minispark_amd.dataframe import DataFrame
    from minispark_amd.sql import Col, Functions, Lit
    from minispark_amd.workloads import api_namespace

    return api_namespace(lambda: DataFrame(object()), Col, Functions, Lit)


# ---- dictionary encoding --------------------------------------------------------------------------------------
WORDS = ["", "a", "AIR", "REG AIR", "RAIL", "x_y", "100%", "MAIL", "aa", "AIRMAIL", "a-long-word-over-seven",
         "another quite long string of thirty+ bytes", "TRUCK", "zz"]


def test_dict_encode_round_trip(engine):
    from minispark_amd import hipspark as hs
    from minispark_amd.constants import ColumnType
    from minispark_amd.io import StrCol

    rng = random.Random(5)
    strings = [rng.choice(WORDS) for _ in range(20_000)]
    raw = StrCol.from_strings(strings)
    dev = engine.dev
    col = dev.upload_raw(raw, ColumnType.STRING)
    coded = dev.dict_encode(col)
    assert coded is not None and coded.kind == hs.STR and coded.fixed_len == 1
    assert list(coded.dict) == sorted({s.encode() for s in strings})  # sorted: codes do not depend on row order
    codes = coded.data[: coded.n].cpu().numpy()
    assert [coded.dict[c].decode() for c in codes.tolist()] == strings
    assert dev.decoded(coded) is col  # table columns keep their plain form
    # decode by gathering from the dictionary (what a column that went through a compaction does)
    import dataclasses

    gathered = dev.decoded(dataclasses.replace(coded, plain=None))
    assert dev.download(gathered, ColumnType.STRING).to_list() == strings
    assert dev.download(coded, ColumnType.STRING).to_list() == strings  # host-side decode of the result hand-over


def test_dict_encode_gives_up_beyond_256_values(engine):
    from minispark_amd.constants import ColumnType
    from minispark_amd.io import StrCol

    dev = engine.dev
    many = dev.upload_raw(StrCol.from_strings([f"key-{i % 300}" for i in range(5000)]), ColumnType.STRING)
    assert dev.dict_encode(many) is None
    exactly = dev.upload_raw(StrCol.from_strings([f"k{i % 256:03d}" for i in range(5000)]), ColumnType.STRING)
    coded = dev.dict_encode(exactly)
    assert coded is not None and len(coded.dict) == 256
    assert dev.download(coded, ColumnType.STRING).to_list() == [f"k{i % 256:03d}" for i in range(5000)]


def _string_table(path, n, seed, blocks=3):
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.io import BlockFile, StrCol

    rng = random.Random(seed)
    s = [rng.choice(WORDS) for _ in range(n)]
    w = [rng.choice(["A", "N", "R"]) for _ in range(n)]
    v = [rng.choice(["REG AIR", "AIR", "RAIL", "SHIP", "TRUCK", "MAIL", "FOB"]) for _ in range(n)]
    i = np.array([rng.randint(-50, 50) for _ in range(n)], np.int32)
    f = np.array([rng.uniform(-10, 10) for _ in range(n)], np.float32)
    schema = [("s", T.STRING), ("w", T.STRING), ("v", T.STRING), ("i", T.INTEGER), ("f", T.FLOAT)]
    cuts = sorted(rng.sample(range(1, n), blocks - 1))
    out = []
    for lo, hi in zip([0, *cuts], [*cuts, n]):
        out.append([StrCol.from_strings(s[lo:hi]), StrCol.from_strings(w[lo:hi]), StrCol.from_strings(v[lo:hi]), i[lo:hi], f[lo:hi]])
    BlockFile(path).write_raw_blocks(schema, out)


DICT_QUERIES = {
    "like_filter_group": lambda api, p: api.DataFrame().table(p).filter(api.Col("v").like("%AIR%"))
        .group_by(api.Col("v")).agg(api.F.sum(api.Col("i")).alias("t"), api.F.count()),
    "concat_key": lambda api, p: api.DataFrame().table(p).filter(api.Col("v").like("%AIR%"))
        .select((api.Col("w") + "-" + api.Col("v")).alias("k"), api.Col("f"), api.Col("i"))
        .group_by(api.Col("k")).agg(api.F.sum(api.Col("f")).alias("sf"), api.F.avg(api.Col("i")).alias("ai"), api.F.count()),
    "compare_literal": lambda api, p: api.DataFrame().table(p).filter((api.Col("s") >= "AIR") & (api.Col("s") != "MAIL"))
        .group_by(api.Col("s")).agg(api.F.count(), api.F.max(api.Col("f")).alias("m")),
    "compare_columns": lambda api, p: api.DataFrame().table(p).filter(api.Col("s") < api.Col("v"))
        .group_by(api.Col("w")).agg(api.F.count()),
    "select_strings": lambda api, p: api.DataFrame().table(p).filter(api.Col("s").like("a%") | (api.Col("v") == "FOB"))
        .select(api.Col("s"), (api.Col("s") + "/" + api.Col("v") + "/" + api.Col("w")).alias("c"), api.Col("i")),
    "like_underscore_and_literal_prefix": lambda api, p: api.DataFrame().table(p).filter(api.Col("s").like("x_y") | api.Col("s").like("100%"))
        .select(api.Col("s"), api.Col("v")),
    "concat_key_after_a_filter_nothing_passes": lambda api, p: api.DataFrame().table(p).filter(api.Col("i") > 1000)
        .select((api.Col("w") + "-" + api.Col("v")).alias("k"), api.Col("f"))
        .group_by(api.Col("k")).agg(api.F.sum(api.Col("f")).alias("sf"), api.F.count()),
    "long_concat_falls_back": lambda api, p: api.DataFrame().table(p)
        .select((api.Col("s") + api.Col("s") + api.Col("v") + api.Col("w")).alias("k"), api.Col("i"))
        .group_by(api.Col("k")).agg(api.F.sum(api.Col("i")).alias("t")),
}


@pytest.mark.parametrize("name", sorted(DICT_QUERIES))
def test_dictionary_coded_queries_match_the_oracle(engine, tmp_path, name):
    from oracle.py_engine import run_query

    path = tmp_path / "t.bin"
    _string_table(path, 6000, seed=11)
    want = run_query(DICT_QUERIES[name](_oracle_api(), str(path)).task)
    frame = DICT_QUERIES[name](_api(engine), str(path))
    for _ in range(3):  # later runs go through the plan / prepared caches
        rows = frame.collect()
        assert_rows_match(rows, want, max_ulps=1)
    table = engine._table(path)
    assert all(table.columns[c].dict is not None for c in table.columns if table.schema[c][0] in ("s", "v")), \
        "string columns with a handful of values must have been dictionary-coded at table open"


def test_dictionary_off_gives_the_same_rows(tmp_path):
    from minispark_amd.execution import HipExecutionEngine
    from oracle.py_engine import run_query

    path = tmp_path / "t.bin"
    _string_table(path, 3000, seed=12)
    want = run_query(DICT_QUERIES["concat_key"](_oracle_api(), str(path)).task)
    with HipExecutionEngine(device=0) as e:
        e.dict_enabled = False
        assert_rows_match(DICT_QUERIES["concat_key"](_api(e), str(path)).collect(), want, max_ulps=1)
        assert all(c.dict is None for c in e._table(path).columns.values())


def test_concatenation_of_dictionaries_that_spell_one_string_twice(engine, tmp_path):
    """dict(a) = {'a', 'ab'}, dict(b) = {'bc', 'c'}: codes (0, 0) and (1, 1) both decode to 'abc'.  A GROUP BY on the
    product code would return 'abc' twice; the reference groups on the string.  Device.dict_concat
    refuses such a product, the plain-string path runs."""
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.io import BlockFile, StrCol
    from oracle.py_engine import run_query

    rng = random.Random(3)
    n = 4000
    a = [rng.choice(["a", "ab"]) for _ in range(n)]
    b = [rng.choice(["bc", "c"]) for _ in range(n)]
    i = np.arange(n, dtype=np.int32)
    path = tmp_path / "c.bin"
    BlockFile(path).write_raw_blocks([("a", T.STRING), ("b", T.STRING), ("i", T.INTEGER)],
                                     [[StrCol.from_strings(a[:1500]), StrCol.from_strings(b[:1500]), i[:1500]],
                                      [StrCol.from_strings(a[1500:]), StrCol.from_strings(b[1500:]), i[1500:]]])

    def query(api):
        C, F = api.Col, api.F
        return (api.DataFrame().table(str(path)).filter(C("i") >= 10).select((C("a") + C("b")).alias("k"), C("i"))
                .group_by(C("k")).agg(F.sum(C("i")).alias("t"), F.count()))

    want = run_query(query(_oracle_api()).task)
    assert sorted(r["k"] for r in want) == ["abbc", "abc", "ac"]
    frame = query(_api(engine))
    for _ in range(3):
        assert_rows_match(frame.collect(), want)
    table = engine._table(path)
    assert all(table.columns[c].dict is not None for c in (0, 1)), "both columns are dictionary-coded at table open"


# ---- the unique-key join --------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["direct", "hashed"])
def test_join_probe_unique_against_numpy(engine, mode):
    import torch

    from minispark_amd import hipspark as hs
    from minispark_amd.device import DCol

    dev = engine.dev
    rng = np.random.default_rng(3)
    n_build, n_probe = 50_000, 200_003
    if mode == "direct":
        keys = rng.permutation(n_build * 4)[:n_build].astype(np.int32) - 1000  # dense range, negative keys included
    else:
        keys = rng.choice(np.arange(-(2**30), 2**30, 977, dtype=np.int64), n_build, replace=False).astype(np.int32)
    probe = np.concatenate([rng.choice(keys, n_probe - 5000), rng.integers(-(2**31), 2**31 - 1, 5000, dtype=np.int64).astype(np.int32)])
    rng.shuffle(probe)
    payload = rng.integers(0, 200, n_build).astype(np.uint8)
    bk, pk = dev.fixed_col(hs.I32, keys), dev.fixed_col(hs.I32, probe)
    pay_col = DCol(hs.STR, dev.to_device(payload, torch.uint8), n_build, lens=dev.const_lens(1, n_build), fixed_len=1)
    dev.reset_flags()
    rows, unit, pay = dev.join_probe_unique(bk, pk, 10, payload=pay_col)
    assert dev.last_join["mode"] == mode
    assert dev.read_flags() == 0
    where = {int(k): i for i, k in enumerate(keys.tolist())}
    want_row = np.array([where.get(int(k), -1) for k in probe.tolist()])
    hit = want_row >= 0
    got_unit = unit.cpu().numpy()
    assert np.array_equal(got_unit == 0xFF, ~hit)
    py_part = np.array([(-2 if int(k) == -1 else int(k)) % 10 for k in probe.tolist()])
    assert np.array_equal(got_unit[hit], py_part[hit].astype(np.uint8))
    assert np.array_equal(rows.cpu().numpy()[hit], want_row[hit]) and not rows.cpu().numpy()[~hit].any()
    assert np.array_equal(pay.cpu().numpy()[hit], payload[want_row[hit]])


@pytest.mark.parametrize("mode", ["direct", "hashed"])
def test_join_build_unique_reports_duplicate_keys(engine, mode):
    from minispark_amd import hipspark as hs

    dev = engine.dev
    keys = np.arange(0, 40_000, dtype=np.int32) * (1 if mode == "direct" else 100_003)
    keys[777] = keys[12_345]
    dev.reset_flags()
    dev.join_probe_unique(dev.fixed_col(hs.I32, keys), dev.fixed_col(hs.I32, keys[:64].copy()), 10)
    assert dev.last_join["mode"] == mode
    assert dev.read_flags() & hs.FLAG_JOIN_DUP


def _join_tables(tmp_path, n_orders, n_li, seed, dup=False, misses=True):
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.io import BlockFile, StrCol
    from minispark_amd.workloads import PRIORITIES, order_key

    rng = np.random.default_rng(seed)
    perm = rng.permutation(n_orders)
    okey = np.array([order_key(int(o)) for o in perm], np.int32)
    if dup:
        okey[5] = okey[100]
    prio = [PRIORITIES[int(c)] for c in rng.integers(0, 5, n_orders)]
    total = np.round(rng.uniform(1, 500, n_orders), 2).astype(np.float32)
    span = int(n_orders * (1.25 if misses else 1.0))  # some lineitems reference orders that do not exist
    lkey = np.array([order_key(int(o)) for o in rng.integers(0, span, n_li)], np.int32)
    qty = rng.integers(1, 51, n_li).astype(np.float32)
    price = (qty * rng.integers(90000, 200001, n_li) / 100.0).astype(np.float32)
    BlockFile(tmp_path / "orders.bin").write_raw_blocks(
        [("o_orderkey", T.INTEGER), ("o_orderpriority", T.STRING), ("o_totalprice", T.FLOAT)],
        [[okey[: n_orders // 2], StrCol.from_strings(prio[: n_orders // 2]), total[: n_orders // 2]],
         [okey[n_orders // 2:], StrCol.from_strings(prio[n_orders // 2:]), total[n_orders // 2:]]])
    third = n_li // 3
    BlockFile(tmp_path / "lineitem.bin").write_raw_blocks(
        [("l_orderkey", T.INTEGER), ("l_quantity", T.FLOAT), ("l_extendedprice", T.FLOAT)],
        [[lkey[:third], qty[:third], price[:third]], [lkey[third: 2 * third], qty[third: 2 * third], price[third: 2 * third]],
         [lkey[2 * third:], qty[2 * third:], price[2 * third:]]])
    return str(tmp_path / "orders.bin"), str(tmp_path / "lineitem.bin")


def _join_queries(api, orders, lineitem):
    C, F = api.Col, api.F
    from minispark_amd import workloads

    def both():
        o = api.DataFrame().table(orders).select(C("o_orderkey"), C("o_orderpriority"), C("o_totalprice"))
        l = api.DataFrame().table(lineitem).select(C("l_orderkey"), C("l_quantity"), C("l_extendedprice"))
        return o.join(l, on=C("o_orderkey") == C("l_orderkey"), how="inner")

    return {
        "config4": workloads.join_group(api, orders, lineitem),
        "filtered_with_build_side_argument": both().filter(C("l_quantity") > 10).group_by(C("o_orderpriority")).agg(
            F.sum(C("o_totalprice") * C("l_quantity")).alias("w"), F.min(C("l_extendedprice")).alias("lo"), F.count()),
        "probe_side_int_key": both().group_by(C("l_orderkey")).agg(F.count(), F.sum(C("l_quantity")).alias("q")),
        "filtered_on_the_probe_side": both().filter((C("l_quantity") > 10) & (C("l_extendedprice") < 60000.0))
            .group_by(C("o_orderpriority")).agg(F.sum(C("l_extendedprice")).alias("rev"), F.avg(C("l_quantity")).alias("aq"), F.count()),
        "count_only": both().group_by(C("o_orderpriority")).agg(F.count()),
    }


JOIN_NAMES = ["config4", "filtered_with_build_side_argument", "probe_side_int_key", "filtered_on_the_probe_side", "count_only"]
# which of them the fused probe (round 3: byte table, probe inside the aggregate) holds; the others - a second build-side
# column - take round 2's materialising in-place join
FUSED_PROBE = {"config4", "probe_side_int_key", "filtered_on_the_probe_side", "count_only"}


@pytest.mark.parametrize("probe", ["fused", "materialised"])
@pytest.mark.parametrize("name", JOIN_NAMES)
def test_in_place_join_matches_the_oracle(engine, tmp_path, name, probe):
    from oracle.py_engine import run_query

    engine.fused_probe_enabled = probe == "fused"
    orders, lineitem = _join_tables(tmp_path, 3000 if name != "probe_side_int_key" else 300, 20_000, seed=21)
    want = run_query(_join_queries(_oracle_api(), orders, lineitem)[name].task)
    frame = _join_queries(_api(engine), orders, lineitem)[name]
    for _ in range(4):  # first run, second (recorded), replays
        flips = assert_rows_match(frame.collect(), want, max_ulps=1)
        assert flips <= 2  # the shared-dictionary tier adds in hardware order
    assert engine.fused_joins >= 1, "the in-place join must have run"
    if probe == "fused" and name in FUSED_PROBE:
        assert engine.fused_probes >= 1 and engine.dev.last_join["mode"] == "byte table" and not engine._no_join8
        assert engine.replays >= 1, "the fused join is recorded and replayed like any query"
    else:
        assert engine.fused_probes == 0 or engine._no_join8 or name not in FUSED_PROBE


@pytest.mark.parametrize("variant", ["plain", "segments", "duplicate", "out_of_range", "no_payload", "wide"])
def test_join8_build_against_numpy(engine, variant):
    """hs_join8_build: window histogram -> scan -> (offset, payload) tuples by window -> windows assembled in LDS.
    The table must equal a plain scatter; duplicate keys, keys outside the table and padded segments are reported /
    skipped."""
    import torch

    from minispark_amd import hipspark as hs

    dev = engine.dev
    rng = np.random.default_rng(11)
    n, key_min = 200_000, -7000
    slots = 5 * hs.JOIN8_WINDOW + 1234  # six windows, the last one partial
    if variant == "wide":  # more windows than the staged scatter's LDS tables hold: tuples go straight to their place
        slots = 4100 * hs.JOIN8_WINDOW + 77
        keys = (rng.choice(slots, n, replace=False) + key_min).astype(np.int32)
    else:
        keys = (rng.permutation(slots)[:n] + key_min).astype(np.int32)
    n_win = -(-slots // hs.JOIN8_WINDOW)
    payload = rng.integers(0, 255, n).astype(np.uint8)
    seg_len, counts = 0, None
    valid = np.ones(n, dtype=bool)
    if variant == "segments":  # three segments of 60 000 rows, the valid prefixes differ
        seg_len, per = 60_000, [60_000, 12_345, 0]
        keys, payload = keys[: 3 * seg_len].copy(), payload[: 3 * seg_len].copy()
        valid = np.concatenate([np.arange(seg_len) < c for c in per])
        keys[~valid] = rng.integers(-(2**31), 2**31 - 1, int((~valid).sum()), dtype=np.int64).astype(np.int32)  # garbage padding
        counts = dev.to_device(np.asarray(per, dtype=np.int64))
        n = 3 * seg_len
    if variant == "duplicate":
        keys[777] = keys[123_456]
    if variant == "out_of_range":
        keys[5] = key_min + slots
        keys[6] = key_min - 1
        valid[5] = valid[6] = False
    d_keys = dev.to_device(keys)
    d_pay = dev.to_device(payload) if variant != "no_payload" else None
    table = torch.zeros(dev.lib.hs_join8_table_bytes(slots) + 64, dtype=torch.uint8, device=dev.device)
    ws = dev.workspace(dev.lib.hs_join8_ws_bytes(n, slots))
    dev.reset_flags()
    hs.check(dev.lib.hs_join8_build(dev.stream, d_keys.data_ptr(), d_pay.data_ptr() if d_pay is not None else None, n, seg_len,
                                    counts.data_ptr() if counts is not None else None, key_min, slots, table.data_ptr(),
                                    ws.data_ptr(), dev.flags.data_ptr()), "hs_join8_build")
    flags = dev.read_flags()
    got = table.cpu().numpy()[: n_win * hs.JOIN8_WINDOW]
    want = np.full(n_win * hs.JOIN8_WINDOW, 0xFF, dtype=np.uint8)
    want[(keys[valid].astype(np.int64) - key_min)] = payload[valid] if variant != "no_payload" else 0
    if variant == "duplicate":
        assert flags & hs.FLAG_JOIN_DUP
        return
    assert flags == (hs.FLAG_BAD_PROGRAM if variant == "out_of_range" else 0)
    assert np.array_equal(got, want)


def test_duplicate_build_keys_fall_back_to_the_general_join(engine, tmp_path):
    from minispark_amd import workloads
    from oracle.py_engine import run_query

    orders, lineitem = _join_tables(tmp_path, 2000, 9000, seed=22, dup=True)
    want = run_query(workloads.join_group(_oracle_api(), orders, lineitem).task)
    frame = workloads.join_group(_api(engine), orders, lineitem)
    for _ in range(2):
        assert_rows_match(frame.collect(), want, max_ulps=1)
    assert engine._no_fused_join, "duplicate keys must have sent the query through the general join"
    assert engine.fused_probes >= 1, "the byte table's build is what noticed them (HS_FLAG_JOIN_DUP)"


# ---- BASELINE configs 4 and 5 at size -------------------------------------------------------------------------
@pytest.mark.parametrize("sf", [1, 10])
def test_config4_join_group_matches_the_c_port(engine, tmp_path, sf):
    from types import SimpleNamespace

    from tools.bench_configs import JoinWorkload

    wl = JoinWorkload(engine, tmp_path, SimpleNamespace(sf=float(sf), config="join"), 0, 1)
    rows = None
    for _ in range(3):
        rows = wl.frame.collect()
    check = wl.full_check(rows)
    assert check["gpu_matches_oracle_full"], check
    assert check["f32_ulp_flips_full"] <= 2
    assert engine.fused_joins >= 1 and engine.dev.last_join["mode"] == "byte table" and engine.replays >= 1
    assert sum(r["n"] for r in rows) == wl.n_li  # every lineitem finds its order


def test_config4_through_the_materialising_probe(tmp_path):
    """Round 2's form of the in-place join (32-bit row table, unit / payload bytes written per row) stays the path for
    joins the byte table does not hold: kept green at sf=1."""
    from types import SimpleNamespace

    from minispark_amd.execution import HipExecutionEngine
    from tools.bench_configs import JoinWorkload

    with HipExecutionEngine(device=0) as e:
        e.fused_probe_enabled = False
        wl = JoinWorkload(e, tmp_path, SimpleNamespace(sf=1.0, config="join"), 0, 1)
        rows = None
        for _ in range(3):
            rows = wl.frame.collect()
        check = wl.full_check(rows)
        assert check["gpu_matches_oracle_full"], check
        assert e.fused_joins >= 1 and e.fused_probes == 0 and e.dev.last_join["mode"] == "direct"


@pytest.mark.parametrize("sf", [1, 10])
def test_config5_strkey_like_matches_the_c_port(engine, tmp_path, sf):
    from types import SimpleNamespace

    from tools.bench_configs import StrKeyWorkload

    wl = StrKeyWorkload(engine, tmp_path, SimpleNamespace(sf=float(sf), config="strkey"), 0, 1)
    rows = None
    for _ in range(3):
        rows = wl.frame.collect()
    check = wl.full_check(rows)
    assert check["gpu_matches_oracle_full"], check
    assert check["f32_ulp_flips_full"] <= 2
    assert wl.config(rows)["dictionary_coded"]["l_shipmode"] is True
    assert sorted(r["k"] for r in rows) == sorted(f"{f}-{m}" for f in "ANR" for m in ("AIR", "REG AIR"))
