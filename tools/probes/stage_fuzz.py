"""Random scan + GROUP BY queries through the stage-level ABI alone (minispark_amd/stage.py) against the Python oracle."""
import os, random, sys, tempfile, time, traceback
from pathlib import Path
os.environ.setdefault("TZ", "UTC"); time.tzset()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from minispark_amd.constants import ColumnType as T
from minispark_amd.dataframe import DataFrame
from minispark_amd.hipspark import HipSparkError
from minispark_amd.io import BlockFile, StrCol
from minispark_amd.sql import Col, Functions as F, Lit
from minispark_amd.stage import NativeEngine, NativeStage, StageUnsupported
from oracle.compare import assert_rows_match
from oracle.py_engine import run_query

first, count = int(sys.argv[1]), int(sys.argv[2])
scratch = Path(tempfile.mkdtemp(prefix="hs_sf_"))
bad = unsupported = 0
with NativeEngine(0) as engine:
    for seed in range(first, first + count):
        rng = random.Random(seed)
        nr = np.random.default_rng(seed)
        n = rng.choice([1, 37, 800, 5000, 30000])
        blocks = rng.choice([1, 2, 5])
        cols = {"k": nr.integers(-3, rng.choice([2, 9, 14]), n).astype(np.int32),
                "c": [rng.choice("ANR") for _ in range(n)],
                "f": nr.normal(0, 100, n).astype(np.float32), "g": nr.uniform(0, 1, n).astype(np.float32),
                "i": nr.integers(-1000, 1000, n).astype(np.int32),
                "t": (nr.integers(0, 3000, n).astype(np.int64) * 86_400_000_000)}
        schema = [("k", T.INTEGER), ("c", T.STRING), ("f", T.FLOAT), ("g", T.FLOAT), ("i", T.INTEGER), ("t", T.TIMESTAMP)]
        cuts = sorted({0, n, *[rng.randrange(0, n + 1) for _ in range(blocks - 1)]})
        path = scratch / f"t{seed}.bin"
        BlockFile(path).write_raw_blocks(schema, [[cols["k"][lo:hi], StrCol.from_strings(cols["c"][lo:hi]), cols["f"][lo:hi],
                                                   cols["g"][lo:hi], cols["i"][lo:hi], cols["t"][lo:hi]] for lo, hi in zip(cuts, cuts[1:])])
        df = DataFrame(object()).table(str(path))
        for _ in range(rng.randint(0, 2)):
            df = df.filter(rng.choice([Col("g") > 0.3, Col("i") % 3 != 0, (Col("f") < 50.0) & (Col("g") <= 0.9),
                                       Col("t") <= "1975-01-01", Col("c") != "N", Col("i") > 5000]))
        pool = [lambda: F.sum(Col("f")), lambda: F.sum(Col("i")), lambda: F.min(Col("f")), lambda: F.max(Col("i")),
                lambda: F.avg(Col("g")), lambda: F.sum(Col("f") * (Lit(1) - Col("g"))), lambda: F.min(Col("i")),
                lambda: F.max(Col("g")), lambda: F.avg(Col("i")), lambda: F.count()]
        aggs = [fn().alias(f"a{j}") if "count" not in repr(fn) else fn() for j, fn in enumerate(rng.sample(pool[:-1], rng.randint(1, 4)))]
        if rng.random() < 0.6:
            aggs.append(F.count())
        key = rng.choice(["k", "c", "k", "t", "m", "m"])
        if key == "m":  # round 3: a SELECT with a computed INTEGER key in front of the GROUP BY (plan version 2)
            df = df.select((Col("i") % rng.choice([7, 97, 700]) - 40).alias("m"), Col("f"), Col("g"), Col("i"))
        q = df.group_by(Col(key)).agg(*aggs)
        want = run_query(q.task)
        try:
            stage = NativeStage(engine, q.task)
        except (StageUnsupported, HipSparkError) as e:
            unsupported += 1
            continue
        try:
            for _ in range(3):
                got = stage.run()
                assert_rows_match(got, want, max_ulps=1)
        except HipSparkError as e:
            if ("on-chip" in str(e) or "LDS" in str(e)) and len(want) > 256:
                unsupported += 1  # more partial rows than the on-chip merge holds: the per-operator ABI's business
            else:
                bad += 1
                print(f"seed {seed}: {type(e).__name__}: {str(e)[:300]}", flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(f"seed {seed}: {type(e).__name__}: {str(e)[:300]}", flush=True)
        finally:
            stage.close()
print(f"seeds {first}..{first + count - 1}: {bad} failures, {unsupported} not of the stage shape")
