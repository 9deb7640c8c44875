"""Debug helper: the randomised many-group queries of tests/test_gpu_shared_tier.py for a seed range on ONE engine,
with a per-seed report of missing / extra / differing groups.  usage: repro_wide.py first last"""
import os
import pathlib
import random
import sys
import tempfile

sys.path.insert(0, os.getcwd())
from minispark_amd.dataframe import DataFrame
from minispark_amd.execution import HipExecutionEngine
from minispark_amd.sql import Col, Functions, Lit
from oracle.py_engine import run_query
from minispark_amd.workloads import api_namespace
from tests.test_gpu_shared_tier import _wide_query, _wide_table

first, last = int(sys.argv[1]), int(sys.argv[2])
with HipExecutionEngine() as engine:
    for seed in range(first, last + 1):
        rng = random.Random(900 + seed)
        path = pathlib.Path(tempfile.mkdtemp()) / "w.bin"
        _wide_table(path, rng.choice([5_000, 20_000, 50_000]), rng.choice([1, 3, 7]), seed)
        want = run_query(_wide_query(random.Random(seed), api_namespace(lambda: DataFrame(object()), Col, Functions, Lit), str(path)).task)
        frame = _wide_query(random.Random(seed), api_namespace(lambda: DataFrame(engine), Col, Functions, Lit), str(path))
        for it in range(2):
            got = frame.collect()
            if not want:
                if got:
                    print(seed, it, "expected no rows, got", len(got), flush=True)
                continue
            key = next(k for k in want[0] if not (k[0] == "a" and k[1:].isdigit()) and k != "count")
            gk = {r[key] for r in got}
            wk = {r[key] for r in want}
            wm = {r[key]: r for r in want}
            bad = [(r, wm[r[key]]) for r in got if r[key] in wm and r != wm[r[key]]]
            ok = len(got) == len(want) and not bad and gk == wk
            if not ok or os.environ.get("VERBOSE"):
                frame.task.explain()
                print(seed, it, "rows", len(got), len(want), "missing", sorted(wk - gk)[:20], "extra", sorted(gk - wk)[:5],
                      "caps", engine._caps, "differing", len(bad), bad[:3], flush=True)
    print("done", flush=True)
