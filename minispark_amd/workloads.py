"""The BASELINE.json workloads as builder functions (bench.py, smoke(), the golden generator and the tests all
build their queries from here, so there is one text per query).

Every function takes an ``api`` namespace (``DataFrame`` factory, ``Col``, ``F`` = Functions, ``Lit``): the same
text then runs against ``mini_spark`` (the reference, in the build container, to make the goldens) and against
``minispark_amd``.

* ``q1``           - configs 2 / 3: the reference's benchmark query (README.md:141-158, examples/benchmark.py:51-68)
* ``join_group``   - config 4: orders JOIN lineitem ON l_orderkey = o_orderkey GROUP BY o_orderpriority
* ``strkey_like``  - config 5: STRING-key GROUP BY (CONCAT) with a LIKE predicate
"""

from __future__ import annotations

from types import SimpleNamespace
from typing import Any

SHIPMODES = ["REG AIR", "AIR", "RAIL", "SHIP", "TRUCK", "MAIL", "FOB"]
PRIORITIES = ["1-URGENT", "2-HIGH", "3-MEDIUM", "4-NOT SPECIFIED", "5-LOW"]
Q1_CUTOFF = "1998-12-01"


def api_namespace(dataframe_cls: Any, col_cls: Any, functions_cls: Any, lit_cls: Any) -> SimpleNamespace:
    return SimpleNamespace(DataFrame=dataframe_cls, Col=col_cls, F=functions_cls, Lit=lit_cls)


def engine_api(engine: Any) -> SimpleNamespace:
    """The namespace bound to one engine of this package."""
    from .dataframe import DataFrame  # noqa: PLC0415
    from .sql import Col, Functions, Lit  # noqa: PLC0415

    return api_namespace(lambda: DataFrame(engine), Col, Functions, Lit)


def order_key(o: int) -> int:
    """Sparse TPC-H-like order keys: 8 used of every 32."""
    return 32 * (o // 8) + (o % 8) + 1


def q1(api: Any, path: str, cutoff: str = Q1_CUTOFF) -> Any:
    C, F, Lit = api.Col, api.F, api.Lit
    disc_price = C("l_extendedprice") * (Lit(1) - C("l_discount"))
    return (
        api.DataFrame().table(path)
        .filter(C("l_shipdate") <= cutoff)
        .group_by(C("l_returnflag"))
        .agg(
            F.sum(C("l_quantity")).alias("sum_qty"),
            F.sum(C("l_extendedprice")).alias("sum_base_price"),
            F.sum(disc_price).alias("sum_disc_price"),
            F.sum(disc_price * (Lit(1) + C("l_tax"))).alias("sum_charge"),
            F.avg(C("l_quantity")).alias("avg_qty"),
            F.avg(C("l_extendedprice")).alias("avg_price"),
            F.avg(C("l_discount")).alias("avg_disc"),
            F.count().alias("count_order"),
        )
    )


def join_group(api: Any, orders_path: str, lineitem_path: str) -> Any:
    """BASELINE config 4.  The build side is ``orders`` (the reference builds its hash table over the LEFT input,
    tasks.py:201-222), the probe side ``lineitem``; aggregates: COUNT, SUM(l_quantity), SUM and MAX of
    l_extendedprice."""
    C, F = api.Col, api.F
    orders = api.DataFrame().table(orders_path).select(C("o_orderkey"), C("o_orderpriority"))
    lineitem = api.DataFrame().table(lineitem_path).select(C("l_orderkey"), C("l_quantity"), C("l_extendedprice"))
    return (
        orders.join(lineitem, on=C("o_orderkey") == C("l_orderkey"), how="inner")
        .group_by(C("o_orderpriority"))
        .agg(F.count().alias("n"), F.sum(C("l_quantity")).alias("qty"), F.sum(C("l_extendedprice")).alias("revenue"),
             F.max(C("l_extendedprice")).alias("max_price"))
    )


def strkey_like(api: Any, lineitem_path: str) -> Any:
    """BASELINE config 5: key = l_returnflag + "-" + l_shipmode, predicate l_shipmode LIKE '%AIR%'."""
    C, F = api.Col, api.F
    return (
        api.DataFrame().table(lineitem_path)
        .filter(C("l_shipmode").like("%AIR%"))
        .select((C("l_returnflag") + "-" + C("l_shipmode")).alias("k"), C("l_quantity"), C("l_discount"))
        .group_by(C("k"))
        .agg(F.sum(C("l_quantity")).alias("qty"), F.avg(C("l_discount")).alias("avg_disc"), F.count())
    )
