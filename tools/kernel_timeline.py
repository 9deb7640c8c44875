"""Kernel timeline of a rocprofv3 --kernel-trace run from its rocpd database.  usage: kernel_timeline.py results.db [last_n]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, d.group_segment_size, s.arch_vgpr_count "
                        f"from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
t0 = rows[0][1]
for n, s, e, g, w, l, v in rows[-int(sys.argv[2]) if len(sys.argv) > 2 else 0:]:
    print(f"{(s - t0) / 1e6:10.3f} ms  {(e - s) / 1e3:9.1f} us  grid {g // w:6d} x {w:4d} lds {l:6d} vgpr {v:3d} {n[:60]}")
