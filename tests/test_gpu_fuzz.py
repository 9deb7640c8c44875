"""Randomised parity: seeded random tables (negative ints, float zeros, empty strings, ragged multi-block
files) x random queries (filters with arithmetic / comparisons / AND / OR / LIKE / string equality, projections
with int / float / string expressions, GROUP BY on int / string / timestamp keys with SUM / MIN / MAX / AVG /
COUNT of random numeric expressions, joins) through HipExecutionEngine vs the Python oracle (the pinned
restatement of the reference).  Exceptions count as results: both sides must raise the same type."""

from __future__ import annotations

import random
from datetime import datetime, timedelta

import numpy as np
import pytest

from tests.conftest import assert_rows_match

pytestmark = pytest.mark.gpu

WORDS = ["", "a", "AIR", "REG AIR", "RAIL", "x_y", "100%", "MAIL", "aa", "AIRMAIL", "a-long-word-over-seven"]
PATTERNS = ["%AIR%", "A%", "%L", "_", "__", "%", "a%a", "REG AIR", "%_y", "100%", "%-%-%"]


def make_table(path, rng: random.Random, n: int, blocks: int):
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.io import BlockFile, StrCol

    base = datetime(2001, 1, 1)
    cols = {
        "i": np.array([rng.randint(-50, 50) for _ in range(n)], dtype=np.int32),
        "j": np.array([rng.choice([-7, -3, -1, 1, 2, 5, 11]) for _ in range(n)], dtype=np.int32),
        "k": np.array([rng.randint(-4, 4) for _ in range(n)], dtype=np.int32),
        "f": np.array([rng.choice([0.0, -0.0, 1.5, -2.25, 1e-3, 123456.78, -0.1]) * rng.choice([1, 1, 3]) for _ in range(n)],
                      dtype=np.float32),
        "g": np.array([rng.uniform(-100, 100) for _ in range(n)], dtype=np.float32),
        "s": [rng.choice(WORDS) for _ in range(n)],
        "w": [rng.choice(WORDS[:5]) for _ in range(n)],
        "t": np.array([int((base + timedelta(days=rng.randint(0, 40))).timestamp() * 1e6) for _ in range(n)], dtype=np.int64),
    }
    schema = [("i", T.INTEGER), ("j", T.INTEGER), ("k", T.INTEGER), ("f", T.FLOAT), ("g", T.FLOAT), ("s", T.STRING),
              ("w", T.STRING), ("t", T.TIMESTAMP)]
    cuts = sorted(rng.sample(range(1, n), min(blocks - 1, n - 1))) if blocks > 1 else []
    bounds = [0, *cuts, n]
    out = []
    for lo, hi in zip(bounds, bounds[1:]):
        out.append([cols["i"][lo:hi], cols["j"][lo:hi], cols["k"][lo:hi], cols["f"][lo:hi], cols["g"][lo:hi],
                    StrCol.from_strings(cols["s"][lo:hi]), StrCol.from_strings(cols["w"][lo:hi]), cols["t"][lo:hi]])
    BlockFile(path).write_raw_blocks(schema, out)


class Gen:
    """Random expression trees over the table above, valid under the reference's type rules."""

    def __init__(self, rng: random.Random, api):
        self.r, self.C, self.Lit = rng, api.Col, api.Lit

    def int_expr(self, d=0):
        r = self.r
        if d >= 2 or r.random() < 0.35:
            return r.choice([self.C("i"), self.C("j"), self.C("k"), self.Lit(r.randint(-9, 9))]) if r.random() < 0.8 else self.C("i")
        op = r.choice("+-*/%")
        a, b = self.int_expr(d + 1), self.int_expr(d + 1)
        if op == "+":
            return a + b
        if op == "-":
            return a - b
        if op == "*":
            return a * b
        if op == "/":
            return a // self.C("j")  # j is never 0
        return a % self.C("j")

    def float_expr(self, d=0):
        r = self.r
        if d >= 2 or r.random() < 0.35:
            return r.choice([self.C("f"), self.C("g")])
        op = r.choice(["+", "-", "*", "/", "mix", "fd", "mod"])
        a = self.float_expr(d + 1)
        if op == "mix":
            return a * self.int_expr(d + 1)  # INT op FLOAT promotes
        if op == "/":
            return a / self.C("j")
        if op == "fd":
            return a // (self.C("j") * 1.5)
        if op == "mod":
            return a % (self.C("j") * 0.75)
        b = self.float_expr(d + 1)
        return a + b if op == "+" else (a - b if op == "-" else a * b)

    def num_expr(self):
        return self.int_expr() if self.r.random() < 0.5 else self.float_expr()

    def cond(self, d=0):
        r = self.r
        kind = r.choice(["icmp", "fcmp", "like", "seq", "ts", "and", "or"]) if d < 2 else r.choice(["icmp", "fcmp", "like", "seq"])
        cmp_ops = [lambda a, b: a < b, lambda a, b: a <= b, lambda a, b: a > b, lambda a, b: a >= b,
                   lambda a, b: a == b, lambda a, b: a != b]
        if kind == "icmp":
            return r.choice(cmp_ops)(self.int_expr(1), self.int_expr(1))
        if kind == "fcmp":
            return r.choice(cmp_ops)(self.float_expr(1), self.float_expr(1))
        if kind == "like":
            return self.C(r.choice(["s", "w"])).like(r.choice(PATTERNS))
        if kind == "seq":
            return r.choice(cmp_ops)(self.C(r.choice(["s", "w"])), r.choice(WORDS))
        if kind == "ts":
            day = (datetime(2001, 1, 1) + timedelta(days=r.randint(0, 40))).strftime("%Y-%m-%d")
            return r.choice(cmp_ops[:4])(self.C("t"), day)
        # the reference types a comparison by its operands and rejects mixed kinds under & / |: combine like kinds
        same = r.choice(["icmp", "fcmp"])
        mk = lambda: r.choice(cmp_ops)(self.int_expr(1), self.int_expr(1)) if same == "icmp" else r.choice(cmp_ops)(self.float_expr(1), self.float_expr(1))
        return (mk() & mk()) if kind == "and" else (mk() | mk())


def random_query(rng: random.Random, api, table: str, table2: str):
    g = Gen(rng, api)
    C, F = api.Col, api.F
    df = api.DataFrame().table(table)
    for _ in range(rng.choice([0, 1, 1, 2])):
        df = df.filter(g.cond())
    shape = rng.choice(["agg", "agg", "agg", "select", "join_agg", "select_then_agg"])
    if shape == "select":
        cols = [C(rng.choice(["i", "s", "t", "f"]))]
        cols.append(g.int_expr().alias("e_int"))
        cols.append(g.float_expr().alias("e_float"))
        if rng.random() < 0.6:
            cols.append((C("s") + rng.choice(["", "-", " / "]) + C("w")).alias("e_str"))
        return df.select(*cols)
    if shape == "join_agg":
        right = api.DataFrame().table(table2).select(C("k").alias("rk"), C("g").alias("rg"), C("w").alias("rw"))
        left = df.select(C("k"), C("i"), C("f"), C("s"))
        joined = left.join(right, on=C("k") == C("rk"), how="inner")
        key = rng.choice(["s", "rw", "k"])
        return joined.group_by(C(key)).agg(F.count(), F.sum(C("f") * C("rg")).alias("x"), F.max(C("i")).alias("m"))
    if shape == "select_then_agg":
        df = df.select((C("w") + "|" + C("s")).alias("key"), C("i"), C("g"))
        return df.group_by(C("key")).agg(F.sum(C("i")).alias("si"), F.avg(C("g")).alias("ag"), F.count())
    key = rng.choice(["k", "w", "s", "t", "i"])
    aggs = []
    for n in range(rng.randint(1, 5)):
        fn = rng.choice([F.sum, F.min, F.max, F.avg, F.sum])
        aggs.append(fn(g.num_expr()).alias(f"a{n}"))
    if rng.random() < 0.7:
        aggs.append(F.count())
    return df.group_by(C(key)).agg(*aggs)


@pytest.fixture(scope="module")
def engine():
    from minispark_amd.execution import HipExecutionEngine

    with HipExecutionEngine(0) as e:
        yield e


_FIRST = int(__import__("os").environ.get("HIPSPARK_FUZZ_FIRST", "0"))


@pytest.mark.parametrize("seed", list(range(_FIRST, _FIRST + int(__import__("os").environ.get("HIPSPARK_FUZZ_SEEDS", "48")))))
def test_random_query_matches_oracle(engine, tmp_path, seed):
    from minispark_amd.dataframe import DataFrame
    from minispark_amd.sql import Col, Functions, Lit
    from oracle.py_engine import run_query
    from tests.queries import api_namespace

    rng = random.Random(1000 + seed)
    n = rng.choice([1, 7, 200, 3000])
    t1, t2 = tmp_path / "a.bin", tmp_path / "b.bin"
    make_table(t1, rng, n, blocks=rng.choice([1, 2, 5]))
    make_table(t2, rng, rng.choice([1, 50, 400]), blocks=rng.choice([1, 3]))

    def run(build_engine):
        api = api_namespace(lambda: DataFrame(build_engine), Col, Functions, Lit)
        return random_query(random.Random(seed), api, str(t1), str(t2))

    try:
        want = run_query(run(object()).task)
        want_err = None
    except Exception as e:  # noqa: BLE001 - the exception type is the expected outcome
        want, want_err = None, type(e).__name__
    frame = run(engine)
    if want_err is not None:
        with pytest.raises(Exception) as info:
            frame.collect()
        assert type(info.value).__name__ == want_err, f"oracle raised {want_err}, engine raised {info.value!r}"
        return
    got = frame.collect()
    assert_rows_match(got, want, max_ulps=1)
    # and through the other route after the scan kernel (short tail <-> general operator sequence)
    engine.short_tail_enabled = not engine.short_tail_enabled
    try:
        assert_rows_match(run(engine).collect(), want, max_ulps=1)
    finally:
        engine.short_tail_enabled = not engine.short_tail_enabled


@pytest.mark.parametrize("seed", [3, 11, 19, 27, 40, 41, 52, 77])
def test_random_query_matches_oracle_on_bigger_tables(engine, tmp_path, seed):
    """The same random queries over 40 000-row tables in 6 blocks: several chunks per unit, dictionaries that grow
    through retries, the shared-dictionary tier, multi-block joins."""
    from minispark_amd.dataframe import DataFrame
    from minispark_amd.sql import Col, Functions, Lit
    from oracle.py_engine import run_query
    from tests.queries import api_namespace

    rng = random.Random(5000 + seed)
    t1, t2 = tmp_path / "a.bin", tmp_path / "b.bin"
    make_table(t1, rng, 40_000, blocks=6)
    make_table(t2, rng, 300, blocks=2)

    def run(build_engine):
        api = api_namespace(lambda: DataFrame(build_engine), Col, Functions, Lit)
        return random_query(random.Random(seed), api, str(t1), str(t2))

    try:
        want = run_query(run(object()).task)
        want_err = None
    except Exception as e:  # noqa: BLE001
        want, want_err = None, type(e).__name__
    frame = run(engine)
    if want_err is not None:
        with pytest.raises(Exception) as info:
            frame.collect()
        assert type(info.value).__name__ == want_err
        return
    for _ in range(2):
        assert_rows_match(frame.collect(), want, max_ulps=1)
