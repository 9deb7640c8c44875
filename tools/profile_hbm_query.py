"""cProfile of the 125 000-group GROUP BY of tools/bench_hbm_tier.py (where does a large-result query spend its host time)."""
import cProfile, os, pstats, sys, tempfile, time
from pathlib import Path
os.environ.setdefault("TZ", "UTC"); time.tzset()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from minispark_amd import constants, hipspark as hs, synth
from minispark_amd.constants import ColumnType as T
from minispark_amd.dataframe import DataFrame
from minispark_amd.device import DCol
from minispark_amd.execution import HipExecutionEngine
from minispark_amd.io import BlockFile
from minispark_amd.sql import Col, Functions as F
from minispark_amd.table import DeviceTable

rows, g = 59_986_052, 500_000
scratch = Path(tempfile.mkdtemp(prefix="hs_hbmp_", dir="/dev/shm"))
constants.SHUFFLE_FOLDER = scratch / "shuffle"
engine = HipExecutionEngine(0)
dev = engine.dev
li = synth.make_lineitem(dev, scratch / "li.bin", rows, with_orderkey=True)
schema = [("k", T.INTEGER), ("v", T.FLOAT)]
path = scratch / "t.bin"
BlockFile(path, schema).write_rows([])
kk = dev.empty(rows, torch.int32); kk.copy_((li.columns[0].data[:rows].to(torch.int64) * 2654435761 % g).to(torch.int32))
engine.attach_device_table(path, DeviceTable(path, schema, synth.block_sizes(rows), {0: DCol(hs.I32, kk, rows), 1: li.columns[2]}, ()))
q = DataFrame(engine).table(str(path)).group_by(Col("k")).agg(F.sum(Col("v")).alias("s"), F.count())
for _ in range(3): q.collect()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): out = q.collect()
pr.disable()
stats = pstats.Stats(pr)
top = sorted(((ct / 5 * 1e3, tt / 5 * 1e3, nc / 5, f"{Path(fn).name}:{line}({fname})") for (fn, line, fname), (cc, nc, tt, ct, _) in stats.stats.items()), reverse=True)[:45]
for ct, tt, nc, what in top:
    print(f"{ct:9.3f} ms cum {tt:9.3f} ms own {nc:9.1f} calls  {what}")
