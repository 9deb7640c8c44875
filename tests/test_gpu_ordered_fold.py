"""The final merge's wave-wide ordered fold (csrc/hs_agg.hip hs_fold_bucket_wave): few groups, MANY units, so that a
group's bucket of partials is long and a wave folds it.  The result must still be the reference's sequential fp64 fold
of the f32 partials in unit order (aggregate.py:71-84) - bit for bit, for sums that round at every step (the kernel has
to notice and fall back to the chain) as for sums that never round (the parallel scan is taken)."""

from __future__ import annotations

import numpy as np
import pytest

from tests.conftest import assert_rows_match

pytestmark = pytest.mark.gpu


def _table(path, blocks, rows_per_block, seed):
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.io import BlockFile

    rng = np.random.default_rng(seed)
    n = blocks * rows_per_block
    k = rng.integers(0, 3, n).astype(np.int32)
    wild = (rng.normal(0, 1, n) * np.exp2(rng.integers(-40, 40, n))).astype(np.float32)  # every add rounds
    tame = rng.integers(-4000, 4000, n).astype(np.float32)  # integers: every partial and every prefix is exact
    zero = np.where(rng.random(n) < 0.5, np.float32(0.0), np.float32(-0.0)).astype(np.float32)
    i = rng.integers(-10**6, 10**6, n).astype(np.int32)
    schema = [("k", T.INTEGER), ("wild", T.FLOAT), ("tame", T.FLOAT), ("zero", T.FLOAT), ("i", T.INTEGER)]
    out = [[c[b * rows_per_block:(b + 1) * rows_per_block] for c in (k, wild, tame, zero, i)] for b in range(blocks)]
    BlockFile(path).write_raw_blocks(schema, out)


def _query(api, path):
    C, F = api.Col, api.F
    return api.DataFrame().table(path).group_by(C("k")).agg(
        F.sum(C("wild")).alias("s_wild"), F.sum(C("tame")).alias("s_tame"), F.avg(C("tame")).alias("a_tame"),
        F.min(C("wild")).alias("mn_wild"), F.max(C("wild")).alias("mx_wild"), F.min(C("zero")).alias("mn_zero"),
        F.max(C("zero")).alias("mx_zero"), F.sum(C("zero")).alias("s_zero"), F.sum(C("i")).alias("s_i"),
        F.min(C("i")).alias("mn_i"), F.max(C("i")).alias("mx_i"), F.count())


@pytest.mark.parametrize(("blocks", "rows_per_block"), [(150, 300), (40, 1000), (333, 64), (600, 40), (30, 900)])  # <= 64 units: a lane per (group, aggregate) runs the chain; more: the lane-group protocol
def test_long_buckets_fold_like_the_reference(tmp_path, blocks, rows_per_block):
    import struct

    from minispark_amd.dataframe import DataFrame
    from minispark_amd.execution import HipExecutionEngine
    from minispark_amd.sql import Col, Functions, Lit
    from oracle.py_engine import run_query
    from tests.queries import api_namespace

    path = tmp_path / "t.bin"
    _table(path, blocks, rows_per_block, blocks)
    want = run_query(_query(api_namespace(lambda: DataFrame(object()), Col, Functions, Lit), str(path)).task)
    with HipExecutionEngine(0) as engine:
        frame = _query(api_namespace(lambda: DataFrame(engine), Col, Functions, Lit), str(path))
        for _run in range(3):
            got = frame.collect()
            assert assert_rows_match(got, want, max_ulps=0) == 0
            # the sign of a zero SUM is defined (Python's sum starts from int 0: never -0.0); MIN / MAX over +-0.0 ties
            # keep the first row's zero in the reference and an arbitrary one in the scan kernel (DESIGN.md divergences)
            for g, w in zip(sorted(got, key=lambda r: r["k"]), sorted(want, key=lambda r: r["k"])):
                assert struct.pack("<f", g["s_zero"]) == struct.pack("<f", w["s_zero"]), (g["s_zero"], w["s_zero"])
