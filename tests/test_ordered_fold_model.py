"""A Python model of the finish kernel's wave-wide ordered fold (csrc/hs_agg.hip hs_fold_bucket_wave): lane chunk sums,
a Hillis-Steele scan over the lanes and every prefix rebuilt from the scan - each addition with its rounding error
(TwoSum).  The kernel takes the scan's total only when every one of those additions was exact; the claim is that the
total then equals the reference's sequential fp64 sum (aggregate.py:71-84) bit for bit.  The model runs the same
sequence of IEEE operations on the CPU and looks for a counterexample; the kernel itself is tested on the GPU
(tests/test_gpu_ordered_fold.py)."""

from __future__ import annotations

import math

import numpy as np


def _add_exact(a: float, b: float) -> tuple[float, bool]:
    s = a + b
    bb = s - a
    return s, (a - (s - bb)) + (b - bb) == 0.0


def _wave_fold(xs: list[float]) -> tuple[float, bool]:
    m = len(xs)
    c = (m + 63) // 64
    ok, incl, chunks = True, [0.0] * 64, []
    for lane in range(64):
        lo = min(lane * c, m)
        hi = min(lo + c, m)
        s = 0.0
        for j in range(lo, hi):
            s, exact = _add_exact(s, xs[j])
            ok &= exact
        incl[lane] = s
        chunks.append((lo, hi))
    d = 1
    while d < 64:
        new = incl[:]
        for lane in range(d, 64):
            new[lane], exact = _add_exact(incl[lane - d], incl[lane])
            ok &= exact
        incl, d = new, d * 2
    for lane, (lo, hi) in enumerate(chunks):
        q = incl[lane - 1] if lane else 0.0
        for j in range(lo, hi):
            q, exact = _add_exact(q, xs[j])
            ok &= exact
    return incl[63], ok


def test_an_all_exact_scan_equals_the_sequential_sum():
    rng = np.random.default_rng(0)
    taken = refused = 0
    for trial in range(3000):
        m = int(rng.integers(1, 400))
        mode = trial % 5
        if mode == 0:
            xs = rng.normal(0, 1e3, m).astype(np.float32)
        elif mode == 1:
            xs = rng.integers(-4000, 4000, m)
        elif mode == 2:
            xs = (rng.normal(0, 1, m) * np.exp2(rng.integers(-40, 40, m))).astype(np.float32)
        elif mode == 3:
            xs = rng.uniform(1e5, 2e5, m).astype(np.float32)  # like Q1's partial sums: similar magnitudes
        else:
            xs = rng.integers(-8, 8, m) * np.exp2(rng.integers(-3, 30, m))
        values = [float(x) for x in np.asarray(xs, dtype=np.float64)]
        total, exact = _wave_fold(values)
        if not exact:
            refused += 1  # the kernel then runs the sequential chain itself
            continue
        taken += 1
        want = 0.0
        for x in values:
            want = want + x
        assert total == want and math.copysign(1.0, total) == math.copysign(1.0, want), (trial, m)
    assert taken > 1500 and refused > 300  # both outcomes are exercised
