"""Planner + lowering (host logic): stage shapes equal the reference's own plans (recorded in the golden
fixtures by running the reference's planner), schema validation raises the reference's errors, and the
lowering of a task tree built by the REAL reference equals the lowering of this package's mirror."""

from __future__ import annotations

import sys
from pathlib import Path

import pytest

from minispark_amd import hipspark as hs
from minispark_amd.constants import ColumnType as T
from minispark_amd.dataframe import DataFrame
from minispark_amd.lowering import LoweringError, ProgramBuilder, lower_aggregate
from minispark_amd.plan import PhysicalPlan
from minispark_amd.sql import Col, Functions as F, Lit
from tests.conftest import load_golden
from tests.queries import CASES, api_namespace

API = api_namespace(lambda: DataFrame(engine=object()), Col, F, Lit)


@pytest.mark.parametrize("case", [c for c in CASES if not c.expect_error], ids=lambda c: c.name)
def test_stage_shapes_equal_the_references_plan(case):
    golden = load_golden(case.name)
    plan = PhysicalPlan.generate_physical_plan(case.build(API, golden["paths"]).task)
    assert [str(s) for s in plan.stages] == golden["stages"]
    assert [[n, str(t)] for n, t in plan.stages[-1].writer.inferred_schema] == golden["schema"]


def test_aggregate_expansion_shape():
    """reference tests/test_plan.py:182-189: [Load, Agg, WriteShuffle] ; [LoadShuffle, Agg(, Project)]."""
    g = load_golden("q1_multiblock")
    from tests.queries import q1

    plan = PhysicalPlan.generate_physical_plan(q1(API, g["paths"]["lineitem"]).task)
    s0, s1 = plan.stages
    assert [type(t).__name__ for t in s0.full_task.task_chain] == ["LoadTableBlockTask", "FilterTask", "AggregateTask",
                                                                 "WriteToShufflePartitions"]
    assert [type(t).__name__ for t in s1.full_task.task_chain] == ["LoadShuffleFilesTask", "AggregateTask", "ProjectTask",
                                                                 "WriteToLocalFileTask"]
    partial = s0.consumers[-1]
    assert [a.name for a in partial.agg_columns] == ["sum_qty", "sum_base_price", "sum_disc_price", "sum_charge",
                                                     "avg_qty_sum", "avg_qty_count", "avg_price_sum", "avg_price_count",
                                                     "avg_disc_sum", "avg_disc_count", "count_order"]
    assert s1.consumers[0].before_shuffle is False


def test_validation_errors():
    g = load_golden("fruits5_load")["paths"]["fruits5"]
    with pytest.raises(ValueError, match="Unknown columns in projection"):
        DataFrame(engine=object()).table(g).select(Col("nope")).schema
    with pytest.raises(ValueError, match="Unknown columns in aggregation"):
        PhysicalPlan.generate_physical_plan(DataFrame(engine=object()).table(g).group_by(Col("x")).agg(F.count()).task)
    with pytest.raises(TypeError, match="Type mismatch"):
        PhysicalPlan.generate_physical_plan(DataFrame(engine=object()).table(g).filter(Col("fruit") == 3).task)
    with pytest.raises(AssertionError):
        DataFrame(engine=object()).table(g).filter(Col("fruit"))  # a filter needs a comparison / LIKE


def test_auto_generated_names():
    assert (Col("quantity") + 3).name == "quantity_add_lit_3"
    assert F.sum(Col("a") * Col("b")).name == "sum_a_mul_b"
    assert F.count().name == "count" and F.avg(Col("x")).name == "avg_x"
    assert Col("s").like("%a_").name == "s_like_%a_" and Col("s").like("%a_").regex == "^.*a.$"
    assert [a.name for a in F.avg(Col("x")).alias("m").expand_avg()] == ["m_sum", "m_count"]


def test_q1_lowering_shares_accumulators_and_orders_slots():
    schema = [("l_quantity", T.FLOAT), ("l_extendedprice", T.FLOAT), ("l_discount", T.FLOAT), ("l_tax", T.FLOAT),
              ("l_returnflag", T.STRING), ("l_shipdate", T.TIMESTAMP)]
    kinds = [hs.F32, hs.F32, hs.F32, hs.F32, hs.STR, hs.I64]
    g = load_golden("q1_multiblock")
    from tests.queries import q1

    st = PhysicalPlan.generate_physical_plan(q1(API, g["paths"]["lineitem"]).task).stages[0]
    low = lower_aggregate(schema, kinds, [st.consumers[0].condition], st.consumers[1].group_by_column,
                          st.consumers[1].agg_columns)
    assert len(low.acc_ops) == 6 and low.agg_to_acc == [0, 1, 2, 3, 0, 4, 1, 4, 5, 4, 4]
    assert low.acc_is_int == [False, False, False, False, True, False]
    assert low.numeric_slots == 6 and low.program.max_depth <= 4
    ops = [w & 0xFF for w in low.program.ins]
    assert ops.count(hs.OP_FILTER) == 1 and ops.count(hs.OP_KEY) == 1 and ops.count(hs.OP_AGG) == 6
    assert ops.index(hs.OP_FILTER) < ops.index(hs.OP_KEY) < ops.index(hs.OP_AGG)
    # every instruction carries the stack depth it executes at
    depth = 0
    for w in low.program.ins:
        op, sp = w & 0xFF, (w >> 8) & 0xFF
        assert sp == depth
        if op in (hs.OP_LD, hs.OP_LIT, hs.OP_LIKE, hs.OP_STRCMP_LIT, hs.OP_STRCMP_COL):
            depth += 1
        elif op in (hs.OP_FILTER, hs.OP_AGG, hs.OP_OUT) or hs.OP_ADD_F <= op <= hs.OP_OR:
            depth -= 1
    assert depth == 0


def test_lowering_type_rules():
    schema = [("i", T.INTEGER), ("f", T.FLOAT), ("s", T.STRING), ("t", T.TIMESTAMP)]
    kinds = [hs.I32, hs.F32, hs.STR, hs.I64]
    def tag(expr):
        return ProgramBuilder(schema, kinds).lower(expr)

    assert tag(Col("i") + Col("i")) == "I" and tag(Col("i") * Col("f")) == "F"
    assert tag(Col("i") / Col("i")) == "F" and tag(Col("i") // 7) == "I"
    assert tag(Col("t") <= "1998-12-01") == "B" and tag(Col("s") == "x") == "B"
    assert tag((Col("i") > 1) & (Col("f") < 2.0)) == "B" and tag(Col("s").like("%x%")) == "B"
    with pytest.raises(LoweringError):
        tag(Col("s") + 1)
    with pytest.raises(TypeError):
        tag(Col("t") < Col("i"))
    b = ProgramBuilder(schema, kinds)
    assert b.string_parts(Col("s") + "-" + Col("s")).parts == [("col", 2), ("lit", b"-"), ("col", 2)]


REFERENCE = Path("/root/reference/src")


@pytest.mark.skipif(not REFERENCE.exists(), reason="the reference is only present in the build container")
@pytest.mark.parametrize("name", ["q1_multiblock", "join_group", "concat_like", "edge_minmax", "e2e_join_group_having"])
def test_reference_objects_lower_to_the_same_programs(name):
    """Drop-in check: plan the query with the REAL reference's DataFrame + planner, lower its task objects
    with minispark_amd.lowering (class-name dispatch) and compare with the mirror's programs byte for byte."""
    sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
    import make_golden

    df, ex, io, sql, tasks = make_golden.import_reference()
    from tests.queries import case_by_name

    case = case_by_name(name)
    golden = load_golden(name)
    ref_api = api_namespace(lambda: df.DataFrame(engine=object.__new__(ex.PythonExecutionEngine)), sql.Col, sql.Functions, sql.Lit)
    import copy

    ref_plan = ex.PythonExecutionEngine().generate_physical_plan(copy.deepcopy(case.build(ref_api, golden["paths"]).task))
    own_plan = PhysicalPlan.generate_physical_plan(case.build(API, golden["paths"]).task)
    assert len(ref_plan.stages) == len(own_plan.stages)
    compared = 0
    for rs, os_ in zip(ref_plan.stages, own_plan.stages):
        for rt, ot in zip(rs.consumers, os_.consumers):
            assert type(rt).__name__ == type(ot).__name__
            if type(rt).__name__ == "AggregateTask" and rt.before_shuffle:
                in_schema = [(n, T[str(t)]) for n, t in rt.parent_task.inferred_schema]
                kinds = [{T.INTEGER: hs.I32, T.FLOAT: hs.F32, T.STRING: hs.STR, T.TIMESTAMP: hs.I64}[t] for _, t in in_schema]
                a = lower_aggregate(in_schema, kinds, [], rt.group_by_column, rt.agg_columns)
                b = lower_aggregate(ot.parent_task.inferred_schema, kinds, [], ot.group_by_column, ot.agg_columns)
                assert a.program.to_bytes() == b.program.to_bytes() and a.agg_to_acc == b.agg_to_acc
                compared += 1
    assert compared >= 1
