"""Per-kernel totals over a time window of a rocprofv3 --kernel-trace rocpd database.
usage: kernel_sums.py results.db [from_ms [to_ms]]   (times relative to the first dispatch)"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
t0 = rows[0][1]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
acc = defaultdict(lambda: [0, 0.0])
first = last = None
for n, s, e in rows:
    t = (s - t0) / 1e6
    if lo <= t <= hi:
        acc[n][0] += 1
        acc[n][1] += (e - s) / 1e3
        first = s if first is None else first
        last = e
print(f"window {lo}..{hi} ms: {(last - first) / 1e6:.3f} ms wall, {sum(v[1] for v in acc.values()) / 1e3:.3f} ms in kernels")
for n, (c, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{us:10.1f} us  {c:5d} x  {n[:90]}")
