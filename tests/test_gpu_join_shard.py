"""The sharded build of the join's byte table on N ranks (round 4; csrc/hs_join.hip): key stripes of the
probe side's blocks, routing of build rows to the ranks whose stripes hold their key, and a table of which only the
windows a rank can reach are assembled.  Reference: both inputs shuffled on the key (plan.py:186-189), one JoinJob per
partition (plan.py:99-109).  The N-rank end-to-end cases are in tests/test_gpu_distributed.py."""

from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture
def engine():
    from minispark_amd.execution import HipExecutionEngine

    with HipExecutionEngine(device=0) as e:
        yield e


def test_key_stripes_per_unit_against_numpy(engine):
    import torch

    from minispark_amd import hipspark as hs

    dev = engine.dev
    rng = np.random.default_rng(5)
    bounds = np.array([0, 1, 1, 70_000, 70_003, 200_000], dtype=np.int64)  # one row, no row, many, three, many
    keys = rng.integers(-(2**31), 2**31 - 1, int(bounds[-1]), dtype=np.int64).astype(np.int32)
    d_keys, d_bounds = dev.to_device(keys), dev.to_device(bounds)
    out = torch.zeros(2 * (len(bounds) - 1), dtype=torch.int32, device=dev.device)
    hs.check(dev.lib.hs_minmax_i32_units(dev.stream, d_keys.data_ptr(), d_bounds.data_ptr(), len(bounds) - 1, out.data_ptr()),
             "hs_minmax_i32_units")
    got = out.cpu().numpy().reshape(-1, 2)
    for u in range(len(bounds) - 1):
        part = keys[bounds[u]: bounds[u + 1]]
        want = (part.min(), part.max()) if len(part) else (2**31 - 1, -(2**31))
        assert tuple(got[u]) == want, u


def _route(dev, keys, codes, smin, smax, owner, world, expect=None):
    import torch

    from minispark_amd import hipspark as hs

    n = len(keys)
    d_keys, d_codes = dev.to_device(keys), dev.to_device(codes)
    d_min, d_max, d_own = dev.to_device(smin), dev.to_device(smax), dev.to_device(owner)
    ws = dev.workspace(dev.lib.hs_join8_route_ws_bytes(n, world))
    start = torch.zeros(world + 1, dtype=torch.int32, device=dev.device)
    hs.check(dev.lib.hs_join8_route_count(dev.stream, d_keys.data_ptr(), n, d_min.data_ptr(), d_max.data_ptr(), d_own.data_ptr(),
                                          len(smin), world, ws.data_ptr(), start.data_ptr()), "hs_join8_route_count")
    starts = [int(v) for v in start.tolist()]
    total = starts[-1]
    out_keys = torch.full((total + 16,), -1, dtype=torch.int32, device=dev.device)
    out_codes = torch.full((total + 16,), 0xEE, dtype=torch.uint8, device=dev.device)
    d_expect = dev.to_device(np.asarray(expect, dtype=np.int32)) if expect is not None else None
    dev.reset_flags()
    hs.check(dev.lib.hs_join8_route(dev.stream, d_keys.data_ptr(), d_codes.data_ptr(), n, d_min.data_ptr(), d_max.data_ptr(),
                                    d_own.data_ptr(), len(smin), world, ws.data_ptr(), start.data_ptr(),
                                    d_expect.data_ptr() if d_expect is not None else None, out_keys.data_ptr(),
                                    out_codes.data_ptr(), total, dev.flags.data_ptr()), "hs_join8_route")
    return starts, out_keys.cpu().numpy(), out_codes.cpu().numpy(), dev.read_flags()


def test_build_rows_are_routed_to_the_owners_of_their_key_stripes(engine):
    """Stripes of 7 blocks on 3 ranks (block b on rank b % 3): neighbours share their boundary key (an order whose lines
    straddle two blocks), two blocks hold one key only, there is a gap no block covers.  Every (row, destination) pair
    must arrive exactly once, grouped by destination; rows in the gap or outside every stripe go nowhere."""
    from minispark_amd import hipspark as hs

    dev = engine.dev
    world = 3
    smin = np.array([-500, 100, 100, 100, 4000, 9000, 9500], dtype=np.int32)
    smax = np.array([100, 100, 100, 3000, 9000, 9500, 20_000], dtype=np.int32)
    owner = (np.arange(7) % world).astype(np.int32)
    rng = np.random.default_rng(9)
    keys = np.concatenate([rng.integers(-1000, 21_000, 50_000), [100, 100, 9000, 9500, 3500, -501, 20_001]]).astype(np.int32)
    codes = rng.integers(0, 255, len(keys)).astype(np.uint8)
    starts, got_keys, got_codes, flags = _route(dev, keys, codes, smin, smax, owner, world)
    assert flags == 0
    want = {d: [] for d in range(world)}
    for k, c in zip(keys.tolist(), codes.tolist()):
        dests = {int(owner[b]) for b in range(7) if smin[b] <= k <= smax[b]}
        for d in dests:
            want[d].append((k, c))
    for d in range(world):
        got = sorted(zip(got_keys[starts[d]: starts[d + 1]].tolist(), got_codes[starts[d]: starts[d + 1]].tolist()))
        assert got == sorted(want[d]), d
    assert starts[-1] == sum(len(v) for v in want.values())
    # split sizes agreed in an earlier run that no longer fit the data: nothing is written, the stale flag is raised
    stale = list(starts)
    stale[1] += 1
    _, k2, _, flags = _route(dev, keys, codes, smin, smax, owner, world, expect=stale)
    assert flags == hs.FLAG_ROUTE_STALE and (k2[: starts[-1]] == -1).all()
    _, k3, _, flags = _route(dev, keys, codes, smin, smax, owner, world, expect=starts)
    assert flags == 0 and sorted(k3[: starts[-1]].tolist()) == sorted(got_keys[: starts[-1]].tolist())


def test_only_the_marked_windows_of_the_table_are_built(engine):
    import torch

    from minispark_amd import hipspark as hs

    dev = engine.dev
    rng = np.random.default_rng(3)
    key_min, slots = 1000, 6 * hs.JOIN8_WINDOW + 99
    n_win = 7
    mask = np.array([1, 0, 1, 1, 0, 0, 1], dtype=np.uint8)
    offs = np.concatenate([rng.choice(hs.JOIN8_WINDOW, 9000, replace=False) + w * hs.JOIN8_WINDOW for w in (0, 2, 3)]
                          + [6 * hs.JOIN8_WINDOW + rng.choice(99, 40, replace=False)])
    keys = (rng.permutation(offs) + key_min).astype(np.int32)
    codes = rng.integers(0, 255, len(keys)).astype(np.uint8)
    table = torch.full((int(dev.lib.hs_join8_table_bytes(slots)) + 64,), 0xAB, dtype=torch.uint8, device=dev.device)
    ws = dev.workspace(dev.lib.hs_join8_ws_bytes(len(keys), slots))
    d_keys, d_codes, d_mask = dev.to_device(keys), dev.to_device(codes), dev.to_device(mask)
    dev.reset_flags()
    hs.check(dev.lib.hs_join8_build_windows(dev.stream, d_keys.data_ptr(), d_codes.data_ptr(), len(keys), key_min, slots,
                                            d_mask.data_ptr(), table.data_ptr(), ws.data_ptr(), dev.flags.data_ptr()),
             "hs_join8_build_windows")
    assert dev.read_flags() == 0
    got = table.cpu().numpy()[: n_win * hs.JOIN8_WINDOW].reshape(n_win, hs.JOIN8_WINDOW)
    want = np.full(n_win * hs.JOIN8_WINDOW, 0xFF, dtype=np.uint8)
    want[keys.astype(np.int64) - key_min] = codes
    want = want.reshape(n_win, hs.JOIN8_WINDOW)
    for w in range(n_win):
        if mask[w]:
            assert np.array_equal(got[w], want[w]), w
        else:
            assert (got[w] == 0xAB).all(), f"window {w} is not this rank's: it must not be touched"
    # a row routed into a window the rank does not build is a routing error, not a silent miss
    bad = np.concatenate([keys, [key_min + hs.JOIN8_WINDOW + 5]]).astype(np.int32)
    d_bad, d_codes2 = dev.to_device(bad), dev.to_device(np.concatenate([codes, [1]]).astype(np.uint8))
    ws = dev.workspace(dev.lib.hs_join8_ws_bytes(len(bad), slots))
    dev.reset_flags()
    hs.check(dev.lib.hs_join8_build_windows(dev.stream, d_bad.data_ptr(), d_codes2.data_ptr(), len(bad), key_min, slots,
                                            d_mask.data_ptr(), table.data_ptr(), ws.data_ptr(), dev.flags.data_ptr()),
             "hs_join8_build_windows")
    assert dev.read_flags() == hs.FLAG_BAD_PROGRAM
