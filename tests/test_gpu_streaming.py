"""Tables beyond HBM: with a tiny residency budget the scan stage runs block range by block range -
partial aggregates accumulate across the ranges, rows of non-aggregating queries are appended to the result BlockFile
with the reference's append-merge rule - and the rows must still be the reference's golden rows."""

from __future__ import annotations

import pytest

from tests.conftest import assert_rows_match, load_golden
from tests.queries import case_by_name

pytestmark = pytest.mark.gpu

CASES = ["q1_multiblock", "q1_ragged_blocks", "q1_selective", "concat_like", "many_groups", "fruits5_filter", "edge_int_key"]


@pytest.mark.parametrize("name", CASES)
def test_streamed_scan_matches_reference(name):
    from minispark_amd.execution import HipExecutionEngine
    from minispark_amd.workloads import engine_api

    golden = load_golden(name)
    with HipExecutionEngine(device=0) as engine:
        engine.hbm_budget = 2048  # bytes: every multi-block golden table streams in several ranges
        frame = case_by_name(name).build(engine_api(engine), golden["paths"])
        for _ in range(2):
            flips = assert_rows_match(frame.collect(), golden["rows"], max_ulps=1)
            assert flips <= (2 if name == "many_groups" else 0)
        if name.startswith("q1") or name in ("concat_like", "many_groups"):
            assert engine.streamed_ranges >= 4, "these tables are far larger than the budget: several ranges per run"


def test_streamed_result_file_follows_the_append_merge_rule(tmp_path):
    """A filter-only query over a 7-block table, streamed: the result BlockFile is written range after range; it must read
    back (with the independent oracle reader) to exactly the rows of the resident run, in blocks the reference's writer
    would have produced (no block longer than ROWS_PER_BLOCK, only the last one short)."""
    import numpy as np

    from minispark_amd import constants
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.execution import HipExecutionEngine
    from minispark_amd.io import BlockFile
    from minispark_amd.workloads import engine_api
    from oracle import blockfile as bfio

    rng = np.random.default_rng(8)
    n = 7000
    blocks = [[rng.integers(-100, 100, 1000).astype(np.int32), rng.uniform(0, 1, 1000).astype(np.float32)] for _ in range(n // 1000)]
    path = tmp_path / "t.bin"
    BlockFile(path).write_raw_blocks([("i", T.INTEGER), ("f", T.FLOAT)], blocks)

    def run(budget, rows_per_block):
        constants.ROWS_PER_BLOCK = rows_per_block
        try:
            with HipExecutionEngine(device=0) as engine:
                engine.hbm_budget = budget
                api = engine_api(engine)
                frame = api.DataFrame().table(str(path)).filter(api.Col("f") > 0.25).select(api.Col("i"), (api.Col("f") * 2).alias("g"))
                results = engine.execute_full_task(frame.task)
                files = [f.file_path for r in results for f in r.output_files]
                schema, blocks_read = bfio.read_blockfile(files[0])
                return engine.streamed_ranges, [len(b[0]) for b in blocks_read], [v for b in blocks_read for v in zip(*b)]
        finally:
            constants.ROWS_PER_BLOCK = 2 * 1024 * 1024

    ranges_r, sizes_r, rows_r = run(None, 2 * 1024 * 1024)
    ranges_s, sizes_s, rows_s = run(4096, 1500)
    assert ranges_r == 0 and ranges_s >= 4
    assert rows_s == rows_r and len(rows_s) > 4000
    assert all(sz == 1500 for sz in sizes_s[:-1]) and 0 < sizes_s[-1] <= 1500  # append-merge: full blocks, then the rest


@pytest.mark.parametrize("name", ["config4", "filtered_on_the_probe_side", "count_only"])
def test_join_streams_a_probe_side_that_does_not_fit(tmp_path, name):
    """Round 4: a join whose probe (right) side exceeds the HBM budget.  The reference streams the right side block by
    block through the build side's hash map (tasks.py:224-240); here the probe side's scan stage is deferred and the join
    stage reads it block range by block range: the byte table is built once, every range leaves RAW per-JoinJob tables,
    and they are added up before the one rounding per JoinJob - so the rows equal the oracle's like a resident run."""
    from minispark_amd.execution import HipExecutionEngine
    from oracle.py_engine import run_query
    from tests.test_gpu_join_dict import _api, _join_queries, _join_tables, _oracle_api

    orders, lineitem = _join_tables(tmp_path, 3000, 24_000, seed=33)
    want = run_query(_join_queries(_oracle_api(), orders, lineitem)[name].task)
    with HipExecutionEngine(device=0) as engine:
        engine.hbm_budget = 150_000  # lineitem's three blocks (96 KB each): two ranges; orders streams through its scan stage
        frame = _join_queries(_api(engine), orders, lineitem)[name]
        for _ in range(2):
            assert assert_rows_match(frame.collect(), want, max_ulps=1) <= 2
        assert engine.streamed_ranges >= 4 and engine.fused_probes >= 2
        assert engine.dev.last_join["mode"] == "byte table"


def test_join_to_the_result_file_streams_its_probe_side(tmp_path):
    """orders JOIN lineitem -> SELECT ... with a probe side beyond the budget: every range's joined rows are appended to
    the result BlockFile (append-merge rule, io.py:231-252); the row multiset equals the oracle's."""
    from minispark_amd.execution import HipExecutionEngine
    from oracle.py_engine import run_query
    from tests.test_gpu_join_dict import _api, _join_tables, _oracle_api

    orders, lineitem = _join_tables(tmp_path, 2000, 9000, seed=34)

    def query(api):
        C = api.Col
        o = api.DataFrame().table(orders).select(C("o_orderkey"), C("o_orderpriority"))
        li = api.DataFrame().table(lineitem).select(C("l_orderkey"), C("l_quantity"), C("l_extendedprice"))
        return (o.join(li, on=C("o_orderkey") == C("l_orderkey"), how="inner").filter(C("l_quantity") > 25)
                .select(C("o_orderpriority"), C("l_orderkey"), (C("l_extendedprice") * 2).alias("twice")))

    want = run_query(query(_oracle_api()).task)
    with HipExecutionEngine(device=0) as engine:
        engine.hbm_budget = 60_000
        rows = query(_api(engine)).collect()
        assert engine.streamed_ranges >= 3
    assert len(rows) == len(want) > 1000
    assert assert_rows_match(rows, want, max_ulps=0) == 0
