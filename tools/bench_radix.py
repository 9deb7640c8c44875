"""Radix tier of GROUP BY alone (hs_group_radix_*): python tools/bench_radix.py [rows] [groups] [units] [reps]
One SUM(f32) over `groups` random int32 keys - the A5 line of tools/bench_ops.py without the rest, for rocprofv3."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", str(Path(__file__).resolve().parent.parent)))
import torch
from minispark_amd import hipspark as hs
from minispark_amd.device import DCol
from minispark_amd.execution import HipExecutionEngine

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 26
G = int(float(sys.argv[2])) if len(sys.argv) > 2 else N // 16
U = int(sys.argv[3]) if len(sys.argv) > 3 else 1
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
engine = HipExecutionEngine(0)
dev = engine.dev
g = torch.Generator(device="cuda"); g.manual_seed(1)
keys = DCol(hs.I32, torch.randint(0, G, (N,), dtype=torch.int32, device="cuda", generator=g), N)
int_keys = keys
STR = int(os.environ.get("RADIX_BENCH_STR", "0"))  # GROUP BY a STRING column of this fixed length: the key's decimal digits
if STR:
    digits = torch.empty((N, STR), dtype=torch.uint8, device="cuda")
    k64 = keys.data[:N].to(torch.int64)
    for i in range(STR):
        digits[:, STR - 1 - i] = ((k64 // 10 ** i) % 10 + 48).to(torch.uint8)
    del k64
    keys = DCol(hs.STR, digits.reshape(-1), N, lens=torch.full((N,), STR, dtype=torch.uint8, device="cuda"), offs=None, fixed_len=STR)
vals = DCol(hs.F32, torch.rand(N, dtype=torch.float32, device="cuda", generator=g), N)
bounds = torch.tensor([N * u // U for u in range(U + 1)], dtype=torch.int64, device="cuda")
biggest = max(N * (u + 1) // U - N * u // U for u in range(U))
COUNT = os.environ.get("RADIX_BENCH_COUNT") == "1"  # a second aggregate: COUNT (a constant 1 that does not travel)
VALUES = [(vals, 0, False)] + ([(None, 1, True)] if COUNT else [])
OPS = [hs.AGG_SUM] * len(VALUES)
for r in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a.record(); out = dev.group_radix(keys, None, N, bounds, U, biggest, VALUES, OPS, True); b.record()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"rows {N} groups {G} units {U}{f' string key of {STR} bytes' if STR else ''}: run {r}: {a.elapsed_time(b):7.3f} ms (host {dt * 1e3:7.3f} ms) = "
          f"{N / a.elapsed_time(b) / 1e6:6.2f} G rows/s, {out[0].n} groups", flush=True)
    if os.environ.get("HIPSPARK_RADIX_STAMPS"):
        import ctypes
        st = (ctypes.c_uint64 * 8)()
        hs.check(dev.lib.hs_group_radix_debug_stamps(st), "stamps")
        names = ["clear", "wait", "slot", "rank", "fold", "emit"]
        waves = max(int(st[6]), 1)
        print("   fold kernel, cycles per wave: " + "  ".join(f"{nm} {int(st[k]) / waves:9.0f}" for k, nm in enumerate(names)) + f"  ({waves} waves)")
# the timed result, checked over ALL rows (checker = plain torch, fp64: order-free, so a tolerance of a few f32 ulps)
if U == 1:
    key_col, accs, _ = out
    if STR:
        d = key_col.data[: key_col.n * STR].reshape(-1, STR).to(torch.int64) - 48
        got_k = sum(d[:, STR - 1 - i] * 10 ** i for i in range(STR))
    else:
        got_k = key_col.data[: key_col.n].to(torch.int64)
    got_s = accs[0].data[: key_col.n].to(torch.float64)
    uniq, inv = torch.unique(int_keys.data[:N].to(torch.int64), return_inverse=True)
    want = torch.zeros(uniq.numel(), dtype=torch.float64, device="cuda").index_add_(0, inv, vals.data[:N].to(torch.float64))
    o = torch.argsort(got_k)
    assert torch.equal(got_k[o], uniq), "group keys differ"
    rel = ((got_s[o] - want).abs() / want.abs().clamp_min(1e-30)).max().item()
    assert rel < 4e-7, rel
    print(f"check over all {N} rows: {uniq.numel()} groups, keys equal, max relative difference of the f32 sums {rel:.2e}")
# size-independent properties of the timed result, any number of units: the groups' counts add up to the row count and
# their sums to the column's sum (fp64 of f32-rounded group sums: relative 1e-6)
key_col, accs, unit_rows = out
total = accs[0].data[: key_col.n].to(torch.float64).sum().item()
want_total = vals.data[:N].to(torch.float64).sum().item()
assert abs(total - want_total) <= 1e-6 * abs(want_total), (total, want_total)
line = f"properties over all {N} rows / {U} units: {key_col.n} groups, sum of group sums {total:.6e} vs column sum {want_total:.6e}"
if COUNT:
    n_counted = int(accs[1].data[: key_col.n].to(torch.int64).sum().item())
    assert n_counted == N, (n_counted, N)
    line += f", counts add up to {n_counted}"
print(line)
engine.__exit__(None, None, None)
