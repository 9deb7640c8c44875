"""The SQL front-end (minispark_amd/parser.py) against the DataFrame API: a text and its API restatement
must build the same task tree.  The expectations follow the reference's parser tests
(/root/reference/tests/test_parser.py: one API chain per text) and its visitor (parser.py:124-162); the
end-to-end texts additionally run on the GPU against the reference's golden rows (test_gpu_sql.py)."""

from __future__ import annotations

import pytest

from minispark_amd.dataframe import DataFrame
from minispark_amd.parser import GroupByError, SemanticError, SqlSyntaxError, parse_sql
from minispark_amd.sql import Col, Functions as F, Lit
from tests.queries import api_namespace, case_by_name
from tests.sql_texts import E2E_SQL


_MIRROR = {"lt": "gt", "le": "ge", "gt": "lt", "ge": "le", "eq": "eq", "ne": "ne"}


def canon(col) -> str:
    """Expression -> text, with `literal <op> column` comparisons turned around.  `col > 100` in SQL builds
    Lit(100) < col (Lit subclasses Col, so Python asks the reflected method first - in the reference too),
    while the API restatements pass plain Python values: same predicate, two spellings."""
    kind = type(col).__name__
    if kind == "BinaryOperatorColumn":
        left, right, op = col.left_side, col.right_side, col.operator.__name__
        if op in _MIRROR and type(left).__name__ == "Lit" and type(right).__name__ != "Lit":
            left, right, op = right, left, _MIRROR[op]
        return f"({canon(left)} {op} {canon(right)})"
    if kind == "AliasColumn":
        return f"{canon(col.original_col)} AS {col.name}"
    if kind == "LikeColumn":
        return f"{canon(col.original_col)} LIKE {col.pattern!r}"
    if kind == "AggCol":
        return f"{col.type}({canon(col.original_col)}) AS {col.name}"
    return str(col)


def render(task) -> list[str]:
    """Task tree -> nested description (type, expressions, aliases), both join sides included."""
    out = []
    node = task
    while node is not None and type(node).__name__ != "VoidTask":
        line = node.describe()
        if type(node).__name__ == "FilterTask":
            line = f"Filter({canon(node.condition)})"
        if type(node).__name__ == "ProjectTask":
            line = "Project(" + ", ".join(canon(c) for c in node.columns) + ")"
        if type(node).__name__ == "AggregateTask":
            line = f"Aggregate({canon(node.group_by_column)}; " + ", ".join(canon(c) for c in node.agg_columns) + ")"
        if type(node).__name__ == "LoadTableBlockTask":
            line += f" alias={node.alias}"
        if type(node).__name__ == "BroadcastHashJoinTask":
            line += " right=" + repr(render(node.right_side_task))
        out.append(line)
        node = node.parent_task
    return out


def T(name="table"):
    return DataFrame(object()).table(name)


CASES = [
    ("SELECT * FROM 'table';", lambda: T().select(Col("*"))),
    ("\n  SELECT * FROM 'table' AS t;\n ", lambda: T().alias("t").select(Col("*"))),
    ("SELECT col_1, col_2, col3 AS col_3, *, col_4 FROM 'table';",
     lambda: T().select(Col("col_1"), Col("col_2"), Col("col3").alias("col_3"), Col("*"), Col("col_4"))),
    ("SELECT t.col FROM 'table' AS t;", lambda: T().alias("t").select(Col("t.col"))),
    ("SELECT col_1 - 5 FROM 'table';", lambda: T().select(Col("col_1") - Lit(5))),
    ("SELECT col_1 - 5 * col_2 / (col_3 + 2) FROM 'table';",
     lambda: T().select(Col("col_1") - Lit(5) * Col("col_2") / (Col("col_3") + Lit(2)))),
    ("SELECT * FROM 'table' WHERE col_1 > 100;", lambda: T().filter(Col("col_1") > Lit(100)).select(Col("*"))),
    ("SELECT * FROM 'table' WHERE col_1 > col_2;", lambda: T().filter(Col("col_1") > Col("col_2")).select(Col("*"))),
    ("SELECT * FROM 'table' WHERE 100 > col_2;", lambda: T().filter(Lit(100) > Col("col_2")).select(Col("*"))),
    ("SELECT * FROM 'table' WHERE (100 > col_2) AND ((col_2 < col_3) OR (col_4 != 20));",
     lambda: T().filter((Lit(100) > Col("col_2")) & ((Col("col_2") < Col("col_3")) | (Col("col_4") != Lit(20))))
     .select(Col("*"))),
    ("SELECT * FROM 'table' WHERE (col_2 * 10 > col_1 + 2);",
     lambda: T().filter(Col("col_2") * Lit(10) > Col("col_1") + Lit(2)).select(Col("*"))),
    ("SELECT * FROM 'table' WHERE col_1 = col_2;", lambda: T().filter(Col("col_1") == Col("col_2")).select(Col("*"))),
    ("SELECT * FROM 'table' WHERE a > 1 AND b < 2 OR c = 3;",
     lambda: T().filter(((Col("a") > Lit(1)) & (Col("b") < Lit(2))) | (Col("c") == Lit(3))).select(Col("*"))),
    ("SELECT col_1, SUM(col_2) FROM 'table' GROUP BY col_1;",
     lambda: T().group_by(Col("col_1")).agg(F.sum(Col("col_2"))).select(Col("col_1"), Col("sum_col_2"))),
    ("SELECT SUM(col_2), MIN(col_3), MAX(col_4), COUNT() FROM 'table' GROUP BY col_1;",
     lambda: T().group_by(Col("col_1")).agg(F.sum(Col("col_2")), F.min(Col("col_3")), F.max(Col("col_4")), F.count())
     .select(Col("sum_col_2"), Col("min_col_3"), Col("max_col_4"), Col("count"))),
    ("SELECT col_1, SUM(col_2 * 2) FROM 'table' GROUP BY col_1;",
     lambda: T().group_by(Col("col_1")).agg(F.sum(Col("col_2") * Lit(2))).select(Col("col_1"), Col("sum_col_2_mul_lit_2"))),
    ("SELECT col_1, SUM(col_2) AS col_3 FROM 'table' GROUP BY col_1;",
     lambda: T().group_by(Col("col_1")).agg(F.sum(Col("col_2")).alias("col_3")).select(Col("col_1"), Col("col_3"))),
    ("SELECT col_1 FROM 'table' WHERE col_1 > col_2 GROUP BY col_1 ;",
     lambda: T().filter(Col("col_1") > Col("col_2")).group_by(Col("col_1")).agg().select(Col("col_1"))),
    ("SELECT col_1, col_2 FROM 'table' JOIN 'table' ON col_1 = col_2;",
     lambda: T().join(T(), on=Col("col_1") == Col("col_2"), how="inner").select(Col("col_1"), Col("col_2"))),
    ("SELECT * FROM 'table' AS t1 JOIN 'table' AS t2 ON col_1 = col_2;",
     lambda: T().alias("t1").join(T().alias("t2"), on=Col("col_1") == Col("col_2"), how="inner").select(Col("*"))),
    ("SELECT * FROM 'a' FULL JOIN 'b' ON x = y INNER JOIN 'c' ON y = z;",
     lambda: T("a").join(T("b"), on=Col("x") == Col("y"), how="inner").join(T("c"), on=Col("y") == Col("z"), how="inner")
     .select(Col("*"))),
    ("SELECT * FROM 'table' WHERE col_1 BETWEEN col_2 AND col_3;",
     lambda: T().filter(Col("col_1").between(Col("col_2"), Col("col_3"))).select(Col("*"))),
    ("SELECT * FROM 'table' WHERE name LIKE 'a%_b';", lambda: T().filter(Col("name").like("a%_b")).select(Col("*"))),
    ("SELECT a FROM 'dir/sub dir/t-1.bin' WHERE a = -3;",
     lambda: T("dir/sub dir/t-1.bin").filter(Col("a") == Lit(-3)).select(Col("a"))),
    ("SELECT SUM (a) AS s FROM 't' GROUP BY a;",  # falls through to the generic function call (parser.py:362-375)
     lambda: T("t").group_by(Col("a")).agg(F.sum(Col("a")).alias("s")).select(Col("s"))),
]


@pytest.mark.parametrize("sql,build", CASES, ids=[c[0].strip()[:60] for c in CASES])
def test_text_builds_the_same_tree_as_the_api(sql, build):
    assert render(parse_sql(sql, object()).task) == render(build().task)


@pytest.mark.parametrize("name", sorted(E2E_SQL))
def test_end_to_end_texts_build_the_catalogue_queries(name):
    case = case_by_name(name)
    paths = {"users": "/data/users.bin", "orders": "/data/orders.bin"}
    api = api_namespace(lambda: DataFrame(object()), Col, F, Lit)
    assert render(parse_sql(E2E_SQL[name].format(**paths), object()).task) == render(case.build(api, paths).task)


def test_rejections():
    for bad in ["SELECT * FROM 'table'",            # the closing ';' is mandatory
                "SELECT * FROM table;",             # table paths are quoted
                "SELECT FROM 't';", "SELECT a, FROM 't';", "SELECT a FROM 't' WHERE ;",
                "SELECT a FROM 't' GROUP a;", "SELECT a FROM 't'; SELECT b FROM 't';", "select a from 't';"]:
        with pytest.raises(SqlSyntaxError):
            parse_sql(bad, object())
    with pytest.raises(GroupByError):
        parse_sql("SELECT a, b FROM 't' GROUP BY a;", object())
    with pytest.raises(SemanticError):
        parse_sql("SELECT a FROM 't' WHERE UPPER(a) = 'X';", object())
    with pytest.raises(NotImplementedError):  # sql.py:44-45
        parse_sql("SELECT a FROM 't' WHERE NOT a = 1;", object())
    with pytest.raises(ValueError):  # numbers are integers, parser.py:349
        parse_sql("SELECT a FROM 't' WHERE a > 1.5;", object())
    with pytest.raises(TypeError):  # GROUP BY takes one column, dataframe.py:64
        parse_sql("SELECT a, b FROM 't' GROUP BY a, b;", object())


def test_having_renames_and_filters_after_the_aggregate():
    df = parse_sql("SELECT k, COUNT() AS n FROM 't' GROUP BY k HAVING SUM(v) > 10 AND COUNT() > 1;", object())
    lines = render(df.task)
    assert lines[0] == "Project(k, n)"
    assert lines[1].startswith("Filter(") and "_having_sum_v" in lines[1] and "_having_count" in lines[1]
    assert "_having_sum_v" in lines[2] and lines[2].startswith("Aggregate(k;")


@pytest.mark.parametrize("name", sorted(E2E_SQL))
def test_end_to_end_texts_give_the_reference_rows_through_the_oracle(name):
    """text -> own parser -> CPU oracle == the rows the real reference produced for the same text."""
    from oracle.py_engine import run_query
    from tests.conftest import assert_rows_match, load_golden

    golden = load_golden(name)
    rows = run_query(parse_sql(E2E_SQL[name].format(**golden["paths"]), object()).task)
    assert assert_rows_match(rows, golden["rows"], max_ulps=0) == 0
