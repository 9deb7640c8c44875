"""Row-set comparison used by every parity check (tests, __graft_entry__.smoke, bench.py's oracle legs).

TEST INFRASTRUCTURE like the rest of oracle/: nothing under minispark_amd/ imports it.
"""

from __future__ import annotations

import math
import struct


def f32(x: float) -> float:
    return struct.unpack("<f", struct.pack("<f", x))[0]


def sort_rows(rows: list[dict]) -> list[dict]:
    return sorted(rows, key=lambda r: tuple((type(v).__name__, v) for v in r.values()))


def f32_ulps(a: float, b: float) -> int:
    """Distance in f32 units in the last place between two f32-representable values."""
    def key(x: float) -> int:
        (i,) = struct.unpack("<i", struct.pack("<f", x))
        return i if i >= 0 else -(i & 0x7FFFFFFF)
    return abs(key(a) - key(b))


def assert_rows_match(got: list[dict], want: list[dict], *, max_ulps: int = 0) -> int:
    """Multiset equality of result rows (row order is unspecified in the reference,
    execution.py:49): ints / strings / datetimes bit-exact; FLOAT columns equal as f32 - or, with
    max_ulps=1, at most one f32 ulp apart (a re-associated fp64 sum can land on the other side of
    an f32 rounding boundary).  Returns the number of values that differed by an ulp."""
    assert len(got) == len(want), f"row count {len(got)} != {len(want)}\n got={got}\nwant={want}"
    flips = 0
    # sort on the non-float columns first, then on the floats rounded (so a 1-ulp difference cannot reorder rows),
    # then on their exact values and signs: rows that tie on everything coarser (say -0.0 and 7.7e-05 next to equal
    # strings and ints) must still pair up the same way in both lists, whatever order the engines emitted them in
    def stable_key(r):
        floats = [v for v in r.values() if type(v) is float]
        return (tuple((type(v).__name__, v) for v in r.values() if type(v) is not float)
                + tuple((round(v, 3),) for v in floats)
                + tuple((v, math.copysign(1.0, v)) if v == v else (math.inf, 0.0) for v in floats))
    for g, w in zip(sorted(got, key=stable_key), sorted(want, key=stable_key)):
        assert list(g.keys()) == list(w.keys()), f"columns {list(g.keys())} != {list(w.keys())}"
        for k in g:
            gv, wv = g[k], w[k]
            assert type(gv) is type(wv), f"{k}: type {type(gv).__name__} != {type(wv).__name__} ({gv!r} vs {wv!r})"
            if type(gv) is float:
                d = f32_ulps(f32(gv), f32(wv))
                assert d <= max_ulps, f"{k}: {gv!r} vs {wv!r} differ by {d} f32 ulps"
                flips += d != 0
            else:
                assert gv == wv, f"{k}: {gv!r} != {wv!r}"
    return flips
