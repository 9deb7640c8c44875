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
