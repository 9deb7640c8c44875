"""Query catalogue shared by the golden generator (runs them on the REAL reference, in the build
container only), the oracle tests and the GPU parity tests.

Every query is a function of an ``api`` namespace exposing ``DataFrame``, ``Col``, ``F`` (Functions)
so that the same text runs against ``mini_spark`` and against ``minispark_amd``.  Tables are described
as seeded generators of Python rows; the generated BlockFiles are committed under tests/golden/ (data
fixtures), so tests never regenerate them through the reference.

The SQL of the reference's own end-to-end tests (/root/reference/tests/test_e2e.py:88-419,
tests/test_execution.py) is restated through the DataFrame API because the SQL parser's dependency
(parsimonious) is not installed here; the mapping follows parser.py:124-162.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from datetime import datetime, timedelta
from types import SimpleNamespace
from typing import Any, Callable

import numpy as np

# ---- tables ------------------------------------------------------------------------------------------

FRUITS = [  # /root/reference/examples/fruit_aggregation.py:11-17
    {"fruit": "apple", "quantity": 3, "color": "red", "price": 1.5},
    {"fruit": "banana", "quantity": 5, "color": "yellow", "price": 1.9},
    {"fruit": "orange", "quantity": 2, "color": "orange", "price": 1.2},
    {"fruit": "orange", "quantity": 4, "color": "orange", "price": 2.2},
]

FRUITS5 = [  # /root/reference/tests/test_execution.py:19-25
    {"fruit": "apple", "quantity": 3, "color": "red"},
    {"fruit": "banana", "quantity": 5, "color": "yellow"},
    {"fruit": "orange", "quantity": 2, "color": "orange"},
    {"fruit": "apple", "quantity": 4, "color": "green"},
    {"fruit": "banana", "quantity": 7, "color": "yellow"},
]

_USERS = [  # /root/reference/tests/test_e2e.py:21-37
    (1, "Alice", "Smith", 25, "USA"), (2, "Bob", "Johnson", 30, "Canada"), (3, "Charlie", "Brown", 22, "USA"),
    (4, "David", "Wilson", 35, "UK"), (5, "Eva", "Davis", 28, "Canada"), (6, "Frank", "Miller", 40, "USA"),
    (7, "Grace", "Taylor", 27, "UK"), (8, "Hank", "Anderson", 32, "USA"), (9, "Ivy", "Thomas", 26, "Canada"),
    (10, "Jack", "Jackson", 24, "USA"), (11, "Kate", "White", 29, "UK"), (12, "Leo", "Harris", 33, "USA"),
    (13, "Mia", "Martin", 31, "Canada"), (14, "Nick", "Thompson", 23, "UK"), (15, "Olivia", "Garcia", 36, "USA"),
]
_ORDERS = [  # /root/reference/tests/test_e2e.py:39-55
    (1, 1, "Laptop", 1, 1200.0, "2025-01-01"), (2, 2, "Mouse", 2, 25.0, "2025-01-05"),
    (3, 3, "Keyboard", 1, 45.0, "2025-02-10"), (4, 1, "Monitor", 2, 300.0, "2025-03-15"),
    (5, 4, "Laptop", 1, 1100.0, "2025-03-20"), (6, 5, "Mouse", 1, 30.0, "2025-04-01"),
    (7, 6, "Keyboard", 2, 50.0, "2025-04-10"), (8, 7, "Monitor", 1, 280.0, "2025-05-05"),
    (9, 8, "Laptop", 1, 1300.0, "2025-05-10"), (10, 9, "Mouse", 3, 27.0, "2025-06-01"),
    (11, 10, "Keyboard", 1, 40.0, "2025-06-15"), (12, 11, "Monitor", 2, 290.0, "2025-07-01"),
    (13, 12, "Laptop", 1, 1250.0, "2025-07-10"), (14, 13, "Mouse", 2, 26.0, "2025-07-15"),
    (15, 14, "Keyboard", 1, 42.0, "2025-08-01"),
]
USERS = [dict(zip(("user_id", "first_name", "last_name", "age", "country"), r)) for r in _USERS]
ORDERS = [
    dict(zip(("order_id", "user_id", "product", "quantity", "price", "order_date"),
             (*r[:5], datetime.fromisoformat(r[5]))))
    for r in _ORDERS
]

from minispark_amd.workloads import PRIORITIES, SHIPMODES, api_namespace, order_key, q1  # noqa: E402,F401
from minispark_amd import workloads as _wl  # noqa: E402


def lineitem_rows(n: int, seed: int) -> list[dict]:
    """TPC-H-shaped lineitem subset (Q1 columns + join key + shipmode), numpy-seeded."""
    rng = np.random.default_rng(seed)
    qty = rng.integers(1, 51, n)
    cents = rng.integers(90000, 200001, n)
    disc = rng.integers(0, 11, n)
    tax = rng.integers(0, 9, n)
    days = rng.integers(0, 2526, n)
    flag = rng.choice(["A", "N", "N", "R"], n)
    mode = rng.integers(0, 7, n)
    base = datetime(1992, 1, 2)
    rows = []
    for i in range(n):
        rows.append({
            "l_orderkey": order_key(i // 4),
            "l_quantity": float(qty[i]),
            "l_extendedprice": float(np.float32(float(qty[i]) * float(cents[i]) / 100.0)),
            "l_discount": float(np.float32(disc[i] / 100.0)),
            "l_tax": float(np.float32(tax[i] / 100.0)),
            "l_returnflag": str(flag[i]),
            "l_shipdate": base + timedelta(days=int(days[i])),
            "l_shipmode": SHIPMODES[int(mode[i])],
        })
    return rows


def orders_rows(n: int, seed: int) -> list[dict]:
    rng = np.random.default_rng(seed)
    perm = rng.permutation(n)
    prio = rng.integers(0, 5, n)
    return [{"o_orderkey": order_key(int(perm[j])), "o_orderpriority": PRIORITIES[int(prio[j])],
             "o_totalprice": float(np.float32(rng.integers(100, 50000) / 7.0))} for j in range(n)]


def edge_rows() -> list[dict]:
    """MIN/MAX identities, negative ints, -1 (hash(-1) == -2), float zeros, a group hit once."""
    return [
        {"k": -1, "g": "x", "i": 2147483000, "f": 1.5e9, "ts": datetime(2001, 1, 1)},
        {"k": -1, "g": "y", "i": -2147483000, "f": -1.0, "ts": datetime(2001, 1, 2)},
        {"k": 7, "g": "x", "i": -5, "f": 0.0, "ts": datetime(1999, 12, 31, 23, 59, 59)},
        {"k": 7, "g": "z", "i": 0, "f": -0.0, "ts": datetime(2001, 1, 3)},
        {"k": -12, "g": "y", "i": 17, "f": 2.5, "ts": datetime(2001, 1, 1)},
        {"k": 7, "g": "x", "i": 600, "f": 2.0e9, "ts": datetime(2030, 6, 1)},
        {"k": 0, "g": "", "i": 1, "f": 1.25, "ts": datetime(1970, 1, 2)},
    ]


@dataclass
class TableSpec:
    name: str
    rows: Callable[[], list[dict]]
    rows_per_block: int = 2 * 1024 * 1024


@dataclass
class Case:
    name: str
    tables: list[TableSpec]
    build: Callable[[Any, dict[str, str]], Any]  # (api, {table name: path}) -> DataFrame
    rows_per_block: int = 2 * 1024 * 1024  # ROWS_PER_BLOCK while the query runs (shuffle files)
    expect_error: str | None = None
    tags: tuple = ()


def _fruit(api, t):
    return api.DataFrame().table(t["fruits"]).group_by(api.Col("fruit")).agg(
        api.F.sum(api.Col("quantity") * api.Col("price")).alias("total_price"))


def _join_group(api, t):
    return _wl.join_group(api, t["orders"], t["lineitem"])


def _concat_like(api, t):
    return _wl.strkey_like(api, t["lineitem"])


def _edge_minmax(api, t):
    C, F = api.Col, api.F
    return api.DataFrame().table(t["edge"]).group_by(C("g")).agg(
        F.min(C("i")).alias("min_i"), F.max(C("i")).alias("max_i"), F.sum(C("i") // 7).alias("sum_fd"),
        F.sum(C("i") % 7).alias("sum_mod"), F.min(C("f")).alias("min_f"), F.max(C("f")).alias("max_f"),
        F.sum(C("f") / 3).alias("sum_div"), F.count())


def _edge_int_key(api, t):
    C, F = api.Col, api.F
    return (api.DataFrame().table(t["edge"]).filter(C("ts") >= "2000-01-01").filter((C("i") != 0) & (C("k") < 100))
            .group_by(C("k")).agg(F.sum(C("i")).alias("s"), F.count()))


def _edge_overflow(api, t):
    C, F = api.Col, api.F
    return api.DataFrame().table(t["edge"]).group_by(C("g")).agg(F.sum(C("i") * 2).alias("s"))


def _edge_divzero(api, t):
    C, F = api.Col, api.F
    return api.DataFrame().table(t["edge"]).group_by(C("g")).agg(F.sum(C("f") / C("i")).alias("s"))


def _many_groups(api, t):
    """Several hundred groups (computed INTEGER key): beyond the per-lane private tables -> the shared-dictionary
    tier."""
    C, F, Lit = api.Col, api.F, api.Lit
    return (api.DataFrame().table(t["lineitem"]).filter(C("l_shipdate") > "1992-03-01")
            .select((C("l_orderkey") % 331 - 100).alias("bucket"), C("l_extendedprice"), C("l_tax"), C("l_orderkey"))
            .group_by(C("bucket"))
            .agg(F.sum(C("l_extendedprice") * (Lit(1) + C("l_tax"))).alias("gross"), F.avg(C("l_tax")).alias("avg_tax"),
                 F.min(C("l_orderkey")).alias("first_order"), F.max(C("l_extendedprice")).alias("max_price"), F.count()))


def _fruits5_count(api, t):
    return api.DataFrame().table(t["fruits5"]).group_by(api.Col("fruit")).agg(api.F.count())


def _fruits5_multi(api, t):
    C, F = api.Col, api.F
    return api.DataFrame().table(t["fruits5"]).group_by(C("fruit")).agg(
        F.count(), F.min(C("quantity")).alias("min"), F.max(C("quantity")).alias("max"),
        F.sum(C("quantity")).alias("sum"))


def _fruits5_join(api, t):
    C = api.Col
    left = api.DataFrame().table(t["fruits5"]).select(C("fruit").alias("fruit_left"), C("color"))
    right = api.DataFrame().table(t["fruits5"]).select(C("fruit").alias("fruit_right"), C("quantity"))
    return left.join(right, on=C("fruit_left") == C("fruit_right"), how="inner")


def _e2e(sel):
    return lambda api, t: sel(api, api.Col, api.F, t)


CASES: list[Case] = [
    Case("fruit", [TableSpec("fruits", lambda: FRUITS)], _fruit),
    Case("q1_multiblock", [TableSpec("lineitem", lambda: lineitem_rows(6000, 20251003), 1024)],
         lambda api, t: q1(api, t["lineitem"]), tags=("q1",)),
    Case("q1_selective", [TableSpec("lineitem", lambda: lineitem_rows(6000, 20251003), 1024)],
         lambda api, t: q1(api, t["lineitem"], "1995-06-17"), tags=("q1",)),
    Case("q1_ragged_blocks", [TableSpec("lineitem", lambda: lineitem_rows(2501, 7), 333)],
         lambda api, t: q1(api, t["lineitem"], "1998-09-02"), tags=("q1",)),
    Case("join_group", [TableSpec("orders", lambda: orders_rows(300, 11), 128),
                        TableSpec("lineitem", lambda: lineitem_rows(1200, 12), 500)], _join_group),
    Case("concat_like", [TableSpec("lineitem", lambda: lineitem_rows(3000, 5), 700)], _concat_like),
    Case("many_groups", [TableSpec("lineitem", lambda: lineitem_rows(5000, 21), 1300)], _many_groups, tags=("many",)),
    Case("edge_minmax", [TableSpec("edge", edge_rows, 3)], _edge_minmax),
    Case("edge_int_key", [TableSpec("edge", edge_rows, 3)], _edge_int_key),
    Case("edge_overflow", [TableSpec("edge", edge_rows, 3)], _edge_overflow, expect_error="OverflowError"),
    Case("edge_divzero", [TableSpec("edge", edge_rows, 3)], _edge_divzero, expect_error="ZeroDivisionError"),
    # /root/reference/tests/test_execution.py
    Case("fruits5_load", [TableSpec("fruits5", lambda: FRUITS5)], lambda api, t: api.DataFrame().table(t["fruits5"])),
    Case("fruits5_select", [TableSpec("fruits5", lambda: FRUITS5)],
         lambda api, t: api.DataFrame().table(t["fruits5"]).select(api.Col("fruit"))),
    Case("fruits5_expr", [TableSpec("fruits5", lambda: FRUITS5)],
         lambda api, t: api.DataFrame().table(t["fruits5"]).select(api.Col("quantity") + 3)),
    Case("fruits5_alias", [TableSpec("fruits5", lambda: FRUITS5)],
         lambda api, t: api.DataFrame().table(t["fruits5"]).select(api.Col("fruit").alias("fruit_name"))),
    Case("fruits5_star", [TableSpec("fruits5", lambda: FRUITS5)],
         lambda api, t: api.DataFrame().table(t["fruits5"]).select(api.Col("*"))),
    Case("fruits5_filter", [TableSpec("fruits5", lambda: FRUITS5)],
         lambda api, t: api.DataFrame().table(t["fruits5"]).filter(api.Col("quantity") > 3)),
    Case("fruits5_count", [TableSpec("fruits5", lambda: FRUITS5)], _fruits5_count),
    Case("fruits5_multi_agg", [TableSpec("fruits5", lambda: FRUITS5)], _fruits5_multi),
    Case("fruits5_self_join", [TableSpec("fruits5", lambda: FRUITS5)], _fruits5_join),
]

_UO = [TableSpec("users", lambda: USERS), TableSpec("orders", lambda: ORDERS)]


def _add_e2e(name: str, fn: Callable) -> None:
    CASES.append(Case(f"e2e_{name}", _UO, _e2e(fn), tags=("e2e",)))


# /root/reference/tests/test_e2e.py:88-419, one entry per SQL text, restated per parser.py:124-162
_add_e2e("select_star", lambda api, C, F, t: api.DataFrame().table(t["users"]).select(C("*")))
_add_e2e("where_eq_str", lambda api, C, F, t: api.DataFrame().table(t["users"]).filter(C("country") == "USA")
         .select(C("first_name"), C("last_name")))
_add_e2e("concat", lambda api, C, F, t: api.DataFrame().table(t["users"])
         .select((C("first_name") + " " + C("last_name")).alias("full_name")))
_add_e2e("int_arith", lambda api, C, F, t: api.DataFrame().table(t["users"])
         .select(C("user_id"), C("age"), (C("age") + 5).alias("age_in_5_years")))
_add_e2e("where_float_gt", lambda api, C, F, t: api.DataFrame().table(t["orders"]).filter(C("price") > 100).select(C("*")))
_add_e2e("int_times_float", lambda api, C, F, t: api.DataFrame().table(t["orders"])
         .select(C("product"), (C("quantity") * C("price")).alias("total_value")))
_add_e2e("between_ts", lambda api, C, F, t: api.DataFrame().table(t["orders"])
         .filter(C("order_date").between("2025-03-01", "2025-06-01")).select(C("*")))
_add_e2e("like", lambda api, C, F, t: api.DataFrame().table(t["orders"]).filter(C("product").like("%top%")).select(C("*")))
_add_e2e("group_count", lambda api, C, F, t: api.DataFrame().table(t["users"]).group_by(C("country"))
         .agg(F.count().alias("user_count")).select(C("country"), C("user_count")))
_add_e2e("group_sum_expr", lambda api, C, F, t: api.DataFrame().table(t["orders"]).group_by(C("user_id"))
         .agg(F.sum(C("quantity") * C("price")).alias("total_spent")).select(C("user_id"), C("total_spent")))
_add_e2e("group_avg_float", lambda api, C, F, t: api.DataFrame().table(t["orders"]).group_by(C("product"))
         .agg(F.avg(C("price")).alias("avg_price")).select(C("product"), C("avg_price")))
_add_e2e("group_avg_int", lambda api, C, F, t: api.DataFrame().table(t["users"]).group_by(C("country"))
         .agg(F.avg(C("age")).alias("avg_age")).select(C("country"), C("avg_age")))
_add_e2e("having_count", lambda api, C, F, t: api.DataFrame().table(t["orders"]).group_by(C("user_id"))
         .agg(F.count().alias("order_count"), F.count().alias("_having_count"))
         .filter(C("_having_count") > 1).select(C("user_id"), C("order_count")))
_add_e2e("join_select", lambda api, C, F, t: api.DataFrame().table(t["users"]).alias("u")
         .join(api.DataFrame().table(t["orders"]).alias("o"), on=C("u.user_id") == C("o.user_id"), how="inner")
         .select(C("u.first_name"), C("o.product")))
_add_e2e("join_group_count", lambda api, C, F, t: api.DataFrame().table(t["users"]).alias("u")
         .join(api.DataFrame().table(t["orders"]).alias("o"), on=C("u.user_id") == C("o.user_id"), how="inner")
         .group_by(C("u.country")).agg(F.count().alias("orders_count")).select(C("u.country"), C("orders_count")))
_add_e2e("join_group_sum", lambda api, C, F, t: api.DataFrame().table(t["users"]).alias("u")
         .join(api.DataFrame().table(t["orders"]).alias("o"), on=C("u.user_id") == C("o.user_id"), how="inner")
         .group_by(C("u.first_name")).agg(F.sum(C("o.quantity") * C("o.price")).alias("spent"))
         .select(C("u.first_name"), C("spent")))
_add_e2e("join_where_float", lambda api, C, F, t: api.DataFrame().table(t["users"]).alias("u")
         .join(api.DataFrame().table(t["orders"]).alias("o"), on=C("u.user_id") == C("o.user_id"), how="inner")
         .filter(C("o.price") > 100).select(C("u.first_name"), C("o.product"), C("o.price")))
_add_e2e("join_where_ts", lambda api, C, F, t: api.DataFrame().table(t["orders"]).alias("o")
         .join(api.DataFrame().table(t["users"]).alias("u"), on=C("u.user_id") == C("o.user_id"), how="inner")
         .filter(C("o.order_date") > "2025-05-01").select(C("u.first_name"), C("o.product"), C("o.order_date")))
_add_e2e("group_sum_max", lambda api, C, F, t: api.DataFrame().table(t["orders"]).group_by(C("product"))
         .agg(F.sum(C("quantity")).alias("total_quantity"), F.max(C("price")).alias("max_price"))
         .select(C("product"), C("total_quantity"), C("max_price")))
_add_e2e("join_group_having", lambda api, C, F, t: api.DataFrame().table(t["users"]).alias("u")
         .join(api.DataFrame().table(t["orders"]).alias("o"), on=C("u.user_id") == C("o.user_id"), how="inner")
         .group_by(C("u.country"))
         .agg(F.count().alias("orders_count"), F.sum(C("o.quantity") * C("o.price")).alias("total_sales"),
              F.sum(C("o.quantity") * C("o.price")).alias("_having_sum_o.quantity_mul_o.price"))
         .filter(C("_having_sum_o.quantity_mul_o.price") > 500)
         .select(C("u.country"), C("orders_count"), C("total_sales")))


def case_by_name(name: str) -> Case:
    for c in CASES:
        if c.name == name:
            return c
    raise KeyError(name)


