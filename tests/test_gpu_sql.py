"""`.sql()` on the HIP engine: the SQL texts of the reference's end-to-end tests, parsed by this package's own
front-end, must return the rows the real reference returned for them (tests/golden/e2e_*.json)."""

from __future__ import annotations

import pytest

from tests.conftest import assert_rows_match, load_golden
from tests.sql_texts import E2E_SQL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from minispark_amd.execution import HipExecutionEngine

    with HipExecutionEngine() as e:
        yield e


@pytest.mark.parametrize("name", sorted(E2E_SQL))
def test_sql_text_matches_the_reference_rows(engine, name):
    golden = load_golden(name)
    rows = engine.sql(E2E_SQL[name].format(**golden["paths"])).collect()
    assert assert_rows_match(rows, golden["rows"], max_ulps=1) == 0


def test_q1_as_sql(engine):
    """The benchmark query as the reference's README states it in SQL (README.md:141-158)."""
    golden = load_golden("q1_multiblock")
    sql = (
        "SELECT l_returnflag, SUM(l_quantity) AS sum_qty, SUM(l_extendedprice) AS sum_base_price, "
        "SUM(l_extendedprice * (1 - l_discount)) AS sum_disc_price, "
        "SUM(l_extendedprice * (1 - l_discount) * (1 + l_tax)) AS sum_charge, AVG(l_quantity) AS avg_qty, "
        "AVG(l_extendedprice) AS avg_price, AVG(l_discount) AS avg_disc, COUNT() AS count_order "
        f"FROM '{golden['paths']['lineitem']}' WHERE l_shipdate <= '1998-12-01' GROUP BY l_returnflag;"
    )
    rows = engine.sql(sql).collect()
    assert assert_rows_match(rows, golden["rows"], max_ulps=1) == 0
