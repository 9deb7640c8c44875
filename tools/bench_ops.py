"""Per-operator microbenchmark against the HBM roofline (8 TB/s): every generic operator of the path at
N rows, timed with HIP events on the launch stream (median of 7), reported as algorithmic GB/s.
Usage: python tools/bench_ops.py [rows=64M]"""
import ctypes as C, os, sys, time
from pathlib import Path
os.environ.setdefault("TZ", "UTC"); time.tzset()
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from minispark_amd import hipspark as hs
from minispark_amd.constants import ColumnType as T
from minispark_amd.device import DBatch, DCol, Device
from minispark_amd.sql import Col, Lit

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 64 * 1024 * 1024
dev = Device(0)
lib = dev.lib
PEAK = 8000.0
g = torch.Generator(device="cuda"); g.manual_seed(1)

def timed(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]

def report(name, ms, nbytes, note=""):
    gbps = nbytes / (ms * 1e-3) / 1e9
    print(f"{name:46s} {ms:8.3f} ms  {gbps:8.1f} GB/s algorithmic  {gbps / PEAK * 100:5.1f}% of HBM peak  {note}")

print(f"rows = {N}")
i32 = torch.randint(-2**31, 2**31 - 1, (N + 64,), dtype=torch.int32, device="cuda", generator=g)[:N]
f32a = torch.rand(N + 64, device="cuda", generator=g)[:N]
f32b = torch.rand(N + 64, device="cuda", generator=g)[:N]
mask = (torch.rand(N, device="cuda", generator=g) < 0.5).to(torch.uint8)
lens = torch.randint(0, 16, (N,), dtype=torch.uint8, device="cuda", generator=g)

# A1 string offsets scan
offs = dev.empty(N + 1, torch.int64); mm = dev.empty(2, torch.int32); ws = dev.workspace(lib.hs_scan_ws_bytes(N))
ms = timed(lambda: lib.hs_str_offsets(dev.stream, lens.data_ptr(), N, offs.data_ptr(), mm.data_ptr(), ws.data_ptr()))
report("A1 hs_str_offsets (u8 lens -> i64 offsets)", ms, N * (1 + 1 + 1 + 8), "reads lens twice (reduce with min/max, down), writes 8 B/row")

# A3 compaction
sel = dev.empty(N, torch.int64); cnt = dev.empty(1, torch.int64)
ms = timed(lambda: lib.hs_compact(dev.stream, mask.data_ptr(), N, sel.data_ptr(), cnt.data_ptr(), ws.data_ptr()))
k = int(cnt.item())
report("A3 hs_compact (50% selectivity)", ms, 2 * N + 8 * k, "reads mask twice, writes 8 B per kept row")
out4 = dev.empty(N, torch.int32)
ms = timed(lambda: lib.hs_gather_fixed(dev.stream, i32.data_ptr(), 4, N, sel.data_ptr(), k, None, out4.data_ptr(), dev.flags.data_ptr()))
report("A3 hs_gather_fixed 4 B (ascending index list)", ms, k * (8 + 4 + 4))
perm = torch.randperm(N, device="cuda", generator=g)
ms = timed(lambda: lib.hs_gather_fixed(dev.stream, i32.data_ptr(), 4, N, perm.data_ptr(), N, None, out4.data_ptr(), dev.flags.data_ptr()))
report("   hs_gather_fixed 4 B (random permutation)", ms, N * (8 + 4 + 4), "random 4 B reads: sector-bound")

# A4 expression evaluation (generic interpreter kernel, one row per lane)
batch = DBatch([("a", T.FLOAT), ("b", T.FLOAT), ("i", T.INTEGER)], [DCol(hs.F32, f32a, N), DCol(hs.F32, f32b, N), DCol(hs.I32, i32, N)], N)
ms = timed(lambda: dev.eval_numeric(batch, [Col("a") * (Lit(1) - Col("b"))]))
report("A4 hs_eval a*(1-b) -> f64 (compiled)", ms, N * (4 + 4 + 8))
ms = timed(lambda: dev.eval_numeric(batch, [(Col("a") > 0.5) & (Col("b") <= 0.25)]))
report("A4 hs_eval (a>0.5)&(b<=0.25) -> mask", ms, N * (4 + 4 + 1))

# A9 hash partition (stable counting sort, P = 10)
keys = DCol(hs.I32, i32, N)
kb = DBatch([("k", T.INTEGER)], [keys], N)
part = dev.empty(N, torch.uint8); k_hs = keys.as_hs()
ms1 = timed(lambda: lib.hs_partition_ids(dev.stream, C.byref(k_hs), None, N, 10, part.data_ptr()))
report("A9 hs_partition_ids (python hash % 10)", ms1, N * (4 + 1))
pperm = dev.empty(N, torch.int64); pstart = dev.empty(11, torch.int64); pws = dev.workspace(lib.hs_partition_ws_bytes(N, 10))
ms2 = timed(lambda: lib.hs_partition_perm(dev.stream, part.data_ptr(), N, 10, pperm.data_ptr(), pstart.data_ptr(), pws.data_ptr()))
report("A9 hs_partition_perm (stable, 10 parts)", ms2, N * (1 + 1 + 8), "reads ids twice, writes 8 B/row")

# A8 hash join: unique build keys (N/4), probe N rows, ~every probe matches once
nb = N // 4
build = torch.randperm(nb, device="cuda", generator=g).to(torch.int32)
probe = torch.randint(0, nb, (N,), dtype=torch.int32, device="cuda", generator=g)
bcol, pcol = DCol(hs.I32, build, nb), DCol(hs.I32, probe, N)
cap = 16
while cap < 2 * nb: cap *= 2
tk = dev.empty(cap, torch.int64); tr = dev.empty(cap, torch.int64); ss = dev.empty(cap + 1, torch.int64); rows = dev.empty(nb, torch.int64)
jws = dev.workspace(lib.hs_join_build_ws_bytes(nb, cap)); b_hs, p_hs = bcol.as_hs(), pcol.as_hs()
ms = timed(lambda: lib.hs_join_build(dev.stream, C.byref(b_hs), nb, cap, tk.data_ptr(), tr.data_ptr(), ss.data_ptr(), rows.data_ptr(), jws.data_ptr(), dev.flags.data_ptr()))
report(f"A8 hs_join_build ({nb} unique keys, cap {cap})", ms, nb * (4 + 8 + 8) + cap * 16, f"{nb / ms / 1e3:.0f} M keys/s")
counts = dev.empty(N, torch.int64)
ms = timed(lambda: lib.hs_join_count(dev.stream, C.byref(b_hs), C.byref(p_hs), N, cap, tk.data_ptr(), tr.data_ptr(), ss.data_ptr(), counts.data_ptr()))
report("A8 hs_join_count (probe)", ms, N * (4 + 8 + 16 + 8), f"{N / ms / 1e3:.0f} M probes/s; random 8+16 B reads of a {cap * 24 / 1e6:.0f} MB table")
ost = dev.empty(N + 1, torch.int64); sws = dev.workspace(lib.hs_scan_ws_bytes(N))
ms = timed(lambda: lib.hs_exclusive_scan_i64(dev.stream, counts.data_ptr(), N, ost.data_ptr(), sws.data_ptr()))
report("A8 hs_exclusive_scan_i64 (output offsets)", ms, N * (8 + 8 + 8))
nout = int(ost[N].item()); ol = dev.empty(nout, torch.int64); orr = dev.empty(nout, torch.int64)
ms = timed(lambda: lib.hs_join_fill(dev.stream, C.byref(b_hs), C.byref(p_hs), N, cap, tk.data_ptr(), tr.data_ptr(), ss.data_ptr(), rows.data_ptr(), ost.data_ptr(), ol.data_ptr(), orr.data_ptr()))
report("A8 hs_join_fill (emit pairs)", ms, N * (4 + 16 + 8 + 16 + 8) + nout * 16, f"{nout} pairs")

# A8, round 4: the same join through the dense-range CSR (hs_join_dense_*: range partition passes + per-partition assembly in
# LDS; probe = two adjacent offsets per row).  Keys 0 .. nb-1 (every slot used) and TPC-H-like sparse keys (8 of every 32).
for label, bkeys, pkeys in (("dense", build, probe),
                            ("sparse 8/32", (32 * (build // 8) + build % 8 + 1).to(torch.int32), (32 * (probe // 8) + probe % 8 + 1).to(torch.int32))):
    lo, hi = int(bkeys.min().item()), int(bkeys.max().item())
    slots = hi - lo + 1
    starts = dev.empty(slots, torch.int32); drows = dev.empty(nb, torch.int32); lcount = dev.empty(nb, torch.int32)
    dws = dev.workspace(lib.hs_join_dense_ws_bytes(nb, slots))
    bk = dev.empty(nb, torch.int32); bk.copy_(bkeys); pk = dev.empty(N, torch.int32); pk.copy_(pkeys)
    ms_b = timed(lambda: lib.hs_join_dense_build(dev.stream, bk.data_ptr(), nb, lo, slots, starts.data_ptr(), drows.data_ptr(), lcount.data_ptr(), dws.data_ptr(), dev.flags.data_ptr()))
    report(f"A8 hs_join_dense_build ({label}, {slots} slots)", ms_b, nb * (4 + 4) + slots * 4, f"{nb / ms_b / 1e3:.0f} M keys/s; keys in, rows + one word per slot out")
    aux = dev.workspace(lib.hs_join_dense_aux_bytes(N))
    ms_c = timed(lambda: lib.hs_join_dense_count(dev.stream, pk.data_ptr(), N, lo, slots, starts.data_ptr(), drows.data_ptr(), lcount.data_ptr(), counts.data_ptr(), aux.data_ptr()))
    report(f"A8 hs_join_dense_count ({label})", ms_c, N * (4 + 8 + 8), f"{N / ms_c / 1e3:.0f} M probes/s; one random 4 B read of a {slots * 4 / 1e6:.0f} MB slot-word array per probe; counts + 8 B aux out")
    lib.hs_exclusive_scan_i64(dev.stream, counts.data_ptr(), N, ost.data_ptr(), sws.data_ptr())
    nout2 = int(ost[N].item()); assert nout2 == nout, (nout2, nout)
    ms_f = timed(lambda: lib.hs_join_dense_fill(dev.stream, N, drows.data_ptr(), aux.data_ptr(), ost.data_ptr(), ol.data_ptr(), orr.data_ptr()))
    report(f"A8 hs_join_dense_fill ({label})", ms_f, N * (8 + 4) + nout * 16, f"{nout} pairs; a stream: offsets + first rows in, pairs out")
    report(f"A8 dense count + fill ({label})", ms_c + ms_f, N * (4 + 8 + 8) + N * (8 + 4) + nout * 16)
    assert dev.read_flags() == 0

# A8, round 4: ANY INTEGER keys through the hashed windows (hs_join_hash_*): the build keys scattered over the whole int32 range
# (a multiplicative scramble of 0 .. nb-1: unique, no structure the window hash could profit from)
hk = ((build.to(torch.int64) * 2654435761 + 12345) & 0xFFFFFFFF).to(torch.int64)
hbk = dev.empty(nb, torch.int32); hbk.copy_(torch.where(hk >= 2**31, hk - 2**32, hk).to(torch.int32))
hp = ((probe.to(torch.int64) * 2654435761 + 12345) & 0xFFFFFFFF).to(torch.int64)
hpk = dev.empty(N, torch.int32); hpk.copy_(torch.where(hp >= 2**31, hp - 2**32, hp).to(torch.int32))
hslots = int(lib.hs_join_hash_slots(nb))
htable = dev.empty(hslots, torch.int64); hrows = dev.empty(nb, torch.int32); hlc = dev.empty(nb, torch.int32)
hws = dev.workspace(lib.hs_join_hash_ws_bytes(nb))
ms_b = timed(lambda: lib.hs_join_hash_build(dev.stream, hbk.data_ptr(), nb, htable.data_ptr(), hrows.data_ptr(), hlc.data_ptr(), hws.data_ptr(), dev.flags.data_ptr()))
report(f"A8 hs_join_hash_build (any int32 keys, {hslots} slots)", ms_b, nb * (4 + 4) + hslots * 8, f"{nb / ms_b / 1e3:.0f} M keys/s; keys in, rows + the 8-byte slots out")
aux = dev.workspace(lib.hs_join_dense_aux_bytes(N))
ms_c = timed(lambda: lib.hs_join_hash_count(dev.stream, hpk.data_ptr(), N, nb, htable.data_ptr(), hrows.data_ptr(), hlc.data_ptr(), counts.data_ptr(), aux.data_ptr()))
report("A8 hs_join_hash_count (any int32 keys)", ms_c, N * (4 + 8 + 8 + 8), f"{N / ms_c / 1e3:.0f} M probes/s; random 8 B reads of a {hslots * 8 / 1e6:.0f} MB table; counts + 8 B aux out")
lib.hs_exclusive_scan_i64(dev.stream, counts.data_ptr(), N, ost.data_ptr(), sws.data_ptr())
nout3 = int(ost[N].item()); assert nout3 == nout, (nout3, nout)
ms_f = timed(lambda: lib.hs_join_dense_fill(dev.stream, N, hrows.data_ptr(), aux.data_ptr(), ost.data_ptr(), ol.data_ptr(), orr.data_ptr()))
report("A8 hs_join_dense_fill (after the hashed count)", ms_f, N * (8 + 4) + nout * 16, f"{nout} pairs")
report("A8 hashed count + fill", ms_c + ms_f, N * (4 + 8 + 8 + 8) + N * (8 + 4) + nout * 16)
assert dev.read_flags() == 0

# A5 global-tier group build + fold (high cardinality)
ng_keys = torch.randint(0, N // 16, (N,), dtype=torch.int32, device="cuda", generator=g)
gcol = DCol(hs.I32, ng_keys, N)
vals = DCol(hs.F32, f32a, N)
def global_agg():
    slot_start, positions, slot_list, ngr = dev._group_build(gcol, None, N)
    dev._group_fold([vals], [hs.AGG_SUM], [False], slot_start, positions, slot_list, ngr, None, quantise=True)
t0 = time.perf_counter(); global_agg(); torch.cuda.synchronize()
ms = timed(global_agg, reps=3)
report(f"A5 global tier build+fold ({N // 16} groups)", ms, N * (4 + 4), f"{N / ms / 1e3:.0f} M rows/s (bit-exact sequential fold)")

# A5 the same through the radix tier (hs_group_radix_*: partition passes + one wave per partition, round 2)
bounds1 = torch.tensor([0, N], dtype=torch.int64, device="cuda")
def radix_agg():
    return dev.group_radix(gcol, None, N, bounds1, 1, N, [(vals, 0, False)], [hs.AGG_SUM], True)
out = radix_agg(); torch.cuda.synchronize()
assert out is not None
ms = timed(radix_agg, reps=3)
report(f"A5 radix tier partition+fold ({N // 16} groups)", ms, N * (4 + 4), f"{N / ms / 1e3:.0f} M rows/s (bit-exact sequential fold; {out[0].n} groups)")
