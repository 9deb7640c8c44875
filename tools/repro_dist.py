"""Debug helper: one multi-rank fuzz seed of tests/test_gpu_distributed.py with a readable diff.
usage: repro_dist.py seed world"""
import json, os, random, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("TZ", "UTC")
from minispark_amd.dataframe import DataFrame
from minispark_amd.sql import Col, Functions, Lit
from oracle.py_engine import run_query
from minispark_amd.workloads import api_namespace
from tests.test_gpu_fuzz import make_table, random_query
from tests.test_gpu_distributed import _free_port, _run_ranks

seed, world = int(sys.argv[1]), int(sys.argv[2])
tmp = Path(tempfile.mkdtemp())
rng = random.Random(7000 + seed)
make_table(tmp / "a.bin", rng, 4000, blocks=5)
make_table(tmp / "b.bin", rng, 200, blocks=3)
api = api_namespace(lambda: DataFrame(object()), Col, Functions, Lit)
frame = random_query(random.Random(seed), api, str(tmp / "a.bin"), str(tmp / "b.bin"))
frame.task.explain()
want = run_query(frame.task)
got = _run_ranks(f"fuzz:{seed}", world, tmp / "rows.json", _free_port(), want)
key = lambda r: json.dumps({k: (v.hex() if isinstance(v, float) else str(v)) for k, v in r.items()}, sort_keys=True)
from collections import Counter
cg, cw = Counter(map(key, got)), Counter(map(key, want))
print("rows", len(got), len(want))
print("only in got :", list((cg - cw).items())[:10])
print("only in want:", list((cw - cg).items())[:10])
