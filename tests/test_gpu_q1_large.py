"""Q1 at sizes the pure-Python oracle cannot reach, against the C restatement (oracle/q1_oracle.c, itself pinned
to the reference's goldens) on the same seeded synthetic rows, plus size-independent properties: the JIT-compiled
and the interpreted kernels agree bit for bit, repeated runs are bit-identical, the result does not depend on the
chunking of the scan, and counts add up."""

from __future__ import annotations

import os
from datetime import datetime

import numpy as np
import pytest

from tests.conftest import assert_rows_match

pytestmark = pytest.mark.gpu

ROWS = 2 * 2_097_152 + 1_234_567  # two full blocks + a ragged one


@pytest.fixture(scope="module")
def setup(tmp_path_factory):
    from minispark_amd import constants, synth
    from minispark_amd.execution import HipExecutionEngine

    root = tmp_path_factory.mktemp("q1large")
    constants.SHUFFLE_FOLDER = root / "shuffle"
    engine = HipExecutionEngine(0)
    path = root / "lineitem.bin"
    table = synth.make_lineitem(engine.dev, path, ROWS)
    engine.attach_device_table(path, table)
    yield engine, str(path), table
    engine.__exit__(None, None, None)


def _frame(engine, path, cutoff):
    from minispark_amd.dataframe import DataFrame
    from minispark_amd.sql import Col, Functions, Lit
    from tests.queries import api_namespace, q1

    return q1(api_namespace(lambda: DataFrame(engine), Col, Functions, Lit), path, cutoff)


def _oracle_rows(table, cutoff):
    from oracle import blockfile as bfio
    from oracle import q1_native

    cols = q1_native.gen(20251003, 0, ROWS)
    return q1_native.run(cols, table.block_rows, bfio.to_us(datetime.fromisoformat(cutoff)), threads=4), cols


@pytest.mark.parametrize("cutoff", ["1998-12-01", "1998-09-02", "1994-03-15", "1991-01-01"])
def test_q1_matches_c_oracle(setup, cutoff):
    engine, path, table = setup
    rows = _frame(engine, path, cutoff).collect()
    want, cols = _oracle_rows(table, cutoff)
    flips = assert_rows_match(rows, want, max_ulps=1)
    assert flips <= 1, f"{flips} f32-ulp flips"
    # counts add up to the number of rows that pass the predicate, computed independently
    from oracle import blockfile as bfio

    passing = int((cols["l_shipdate"] <= bfio.to_us(datetime.fromisoformat(cutoff))).sum())
    assert sum(r["count_order"] for r in rows) == passing
    if cutoff == "1991-01-01":
        assert rows == []  # nothing survives: no groups, no result file


def test_jit_and_interpreter_agree_bit_for_bit_and_runs_repeat(setup):
    from minispark_amd import hipspark as hs

    engine, path, _ = setup
    lib = hs.load_library()
    frame = _frame(engine, path, "1998-09-02")
    stats0 = (hs.C.c_int32 * 3)() if hasattr(hs, "C") else None
    lib.hs_jit_set_enabled(1)
    jit_rows = frame.collect()
    again = frame.collect()
    assert jit_rows == again, "two runs of the same query must be bit-identical"
    counters = (__import__("ctypes").c_int32 * 3)()
    lib.hs_jit_stats(counters)
    assert counters[1] > 0 and counters[2] == 0, "the JIT-compiled kernel must actually have run"
    lib.hs_jit_set_enabled(0)
    try:
        engine.dev._partial_prepared.clear()
        interp_rows = frame.collect()
    finally:
        lib.hs_jit_set_enabled(1)
    assert jit_rows == interp_rows, "same skeleton, same operator definitions: identical bits"


def test_result_independent_of_scan_chunking(setup, monkeypatch):
    engine, path, _ = setup
    frame = _frame(engine, path, "1998-12-01")
    base = frame.collect()
    for steps in ("1", "4", "32"):
        monkeypatch.setenv("HIPSPARK_CHUNK_STEPS", steps)
        engine.dev._partial_prepared.clear()
        rows = frame.collect()
        assert [r["count_order"] for r in rows] == [r["count_order"] for r in base]
        assert_rows_match(rows, base, max_ulps=1)
    monkeypatch.delenv("HIPSPARK_CHUNK_STEPS")
    engine.dev._partial_prepared.clear()


def test_high_cardinality_group_by_uses_global_tier_and_is_exact(tmp_path):
    """GROUP BY on a key with ~40 000 distinct values per block (far beyond the on-chip tiers): the engine
    falls over to the HBM dictionary + ordered per-group fold.  One lane folds a group's rows front to back,
    i.e. in the reference's own order, so the result must equal the Python oracle BIT FOR BIT (0 flips),
    integers and floats alike."""
    from minispark_amd import constants
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.dataframe import DataFrame
    from minispark_amd.execution import HipExecutionEngine
    from minispark_amd.io import BlockFile
    from minispark_amd.sql import Col, Functions as F
    from oracle.py_engine import run_query

    constants.SHUFFLE_FOLDER = tmp_path / "shuffle"
    rng = np.random.default_rng(5)
    n = 120_000
    keys = rng.integers(-20_000, 20_000, n).astype(np.int32)
    qty = rng.integers(1, 50, n).astype(np.int32)
    price = (rng.integers(100, 100_000, n) / 7.0).astype(np.float32)
    path = tmp_path / "t.bin"
    schema = [("k", T.INTEGER), ("q", T.INTEGER), ("p", T.FLOAT)]
    per = 50_000
    BlockFile(path).write_raw_blocks(schema, [[keys[i: i + per], qty[i: i + per], price[i: i + per]] for i in range(0, n, per)])

    def build(engine):
        return (DataFrame(engine).table(str(path)).filter(Col("q") > 3).group_by(Col("k"))
                .agg(F.sum(Col("q") * Col("p")).alias("rev"), F.count(), F.min(Col("p")).alias("lo"),
                     F.avg(Col("p")).alias("mean"), F.max(Col("q")).alias("hi")))

    with HipExecutionEngine(0) as engine:
        rows = build(engine).collect()
        assert engine._global_partial and engine._global_merge, "the global tier must have been used"
        again = build(engine).collect()
    want = run_query(build(object()).task)
    assert len(rows) == len(want) > 30_000
    assert_rows_match(rows, want, max_ulps=0)
    assert_rows_match(again, want, max_ulps=0)


def test_replay_of_recorded_query_is_identical(setup):
    """The third and later runs of a cached plan replay the recorded device calls: same bits, and a query with
    other literals (different program constants) must not reuse it."""
    engine, path, _ = setup
    frame = _frame(engine, path, "1997-01-01")
    before = engine.replays
    runs = [frame.collect() for _ in range(5)]
    assert all(r == runs[0] for r in runs)
    assert engine.replays - before >= 2, "later runs must have taken the replay path"
    other = _frame(engine, path, "1996-01-01").collect()
    assert other != runs[0]
    assert sum(r["count_order"] for r in other) < sum(r["count_order"] for r in runs[0])


def test_trace_file_holds_query_stage_replay_and_kernel_slices(tmp_path):
    """HipExecutionEngine(trace_file=...) writes a Chrome-trace JSON (opens in ui.perfetto.dev): host spans per
    query / stage / replay and the scan kernel's HIP-event duration on a GPU track."""
    import json

    from minispark_amd import synth
    from minispark_amd.dataframe import DataFrame
    from minispark_amd.execution import HipExecutionEngine
    from minispark_amd.sql import Col, Functions, Lit
    from tests.queries import api_namespace, q1

    trace = tmp_path / "trace.json"
    with HipExecutionEngine(trace_file=trace) as engine:
        path = tmp_path / "li.bin"
        table = synth.make_lineitem(engine.dev, path, 300_000, rows_per_block=65_536)
        engine.attach_device_table(path, table)
        frame = q1(api_namespace(lambda: DataFrame(engine), Col, Functions, Lit), str(path))
        for _ in range(4):
            assert len(frame.collect()) == 3
    events = json.loads(trace.read_text())["traceEvents"]
    names = [e["name"] for e in events if e["ph"] == "X"]
    assert names.count("query") == 4 and any(n.startswith("stage 0: LoadTableBlockTask") for n in names)
    assert any(n.startswith("replay") for n in names)
    kernels = [e for e in events if e["ph"] == "X" and e["name"].startswith("scan kernel")]
    assert len(kernels) == 4 and all(0 < e["dur"] < 1e5 and e["args"]["rows"] == 300_000 for e in kernels)
    # every launch of every run (first runs and captured replays) as its own GPU slice: scan kernel + finish launch
    track_names = {e["tid"]: e["args"]["name"] for e in events if e["ph"] == "M"}
    gpu = [e for e in events if e["ph"] == "X" and "every launch" in track_names.get(e["tid"], "")]
    scans = [e for e in gpu if e["name"].startswith(("k_agg_jit", "void k_agg_main"))]
    finishes = [e for e in gpu if "k_agg_finish" in e["name"]]
    assert len(scans) == 4 and len(finishes) == 4 and all(e["dur"] > 0 for e in scans + finishes)
    for sc, fi in zip(scans, finishes):
        assert fi["ts"] >= sc["ts"] + sc["dur"] - 1e-3  # in stream order, not overlapping


def test_rocprof_kernel_trace_merges_into_a_trace(tmp_path):
    """Tracer.add_rocprof_kernel_trace: a rocprofv3 --kernel-trace CSV becomes one more GPU track."""
    import json

    from minispark_amd.tracing import Tracer

    csv_path = tmp_path / "kernel_trace.csv"
    csv_path.write_text("Kind,Agent_Id,Queue_Id,Kernel_Id,Kernel_Name,Start_Timestamp,End_Timestamp,Grid_Size_X,Workgroup_Size_X\n"
                        "KERNEL_DISPATCH,1,1,7,k_agg_jit,1000000,1300000,192512,256\n"
                        "KERNEL_DISPATCH,1,1,9,void k_agg_finish<true>(AggFinishArgs),1305000,1326000,256,256\n")
    tr = Tracer()
    tr.start("query")
    tr.end()
    assert tr.add_rocprof_kernel_trace(csv_path) == 2
    tr.save(tmp_path / "t.json")
    events = json.loads((tmp_path / "t.json").read_text())["traceEvents"]
    merged = [e for e in events if e["ph"] == "X" and e["name"].startswith(("k_agg_jit", "void k_agg_finish"))]
    # ts are microseconds since the epoch as floats (~1.8e15: a quarter of a microsecond of resolution)
    assert [round(e["dur"]) for e in merged] == [300, 21] and merged[1]["ts"] - merged[0]["ts"] == pytest.approx(305.0, abs=1.0)


def test_capacity_hints_are_per_query_shape(tmp_path):
    """A GROUP BY with many groups grows ITS dictionary capacities; Q1 (3 groups) on the same engine keeps the
    small, fast geometry (4 slots per workgroup, 256-lane workgroups)."""
    from minispark_amd import synth
    from minispark_amd.dataframe import DataFrame
    from minispark_amd.execution import HipExecutionEngine
    from minispark_amd.sql import Col, Functions, Lit
    from tests.queries import api_namespace, q1

    with HipExecutionEngine() as engine:
        path = tmp_path / "li.bin"
        table = synth.make_lineitem(engine.dev, path, 400_000, rows_per_block=100_000)
        engine.attach_device_table(path, table)
        api = api_namespace(lambda: DataFrame(engine), Col, Functions, Lit)
        many = DataFrame(engine).table(str(path)).group_by(Col("l_quantity")).agg(Functions.count())
        assert len(many.collect()) == 50  # l_quantity = 1..50
        assert engine.dev.last_group_cap >= 64
        rows = q1(api, str(path)).collect()
        assert len(rows) == 3
        assert engine.dev.last_scan["group_cap"] == 4 and engine.dev.last_scan["wg_threads"] == 256
        assert len(many.collect()) == 50 and engine.dev.last_group_cap >= 64  # and its own hint is remembered


def test_compiled_programs_are_kept_on_disk_between_processes(tmp_path):
    """A second process asking for the same programs loads their code objects from $HIPSPARK_JIT_CACHE instead of
    compiling them again (hs_jit_disk_hits), gives the same rows, and a damaged cache file is ignored and rewritten."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    script = (
        "import json, sys, time\n"
        f"sys.path.insert(0, {str(root)!r})\n"
        "from minispark_amd import hipspark as hs, synth, workloads\n"
        "from minispark_amd.execution import HipExecutionEngine\n"
        "with HipExecutionEngine(0) as e:\n"
        "    path = sys.argv[1]\n"
        "    e.attach_device_table(path, synth.make_lineitem(e.dev, path, 300_000))\n"
        "    t0 = time.perf_counter()\n"
        "    rows = workloads.q1(workloads.engine_api(e), path).collect()\n"
        "    dt = time.perf_counter() - t0\n"
        "    c = (hs.C.c_int32 * 3)()\n"
        "    e.dev.lib.hs_jit_stats(c)\n"
        "    print(json.dumps({'rows': [[str(v) for v in r.values()] for r in rows], 'hits': e.dev.lib.hs_jit_disk_hits(),\n"
        "                      'compiled': c[0], 'failures': c[2], 'first_query_s': dt}))\n")
    cache = tmp_path / "jit"
    env = dict(os.environ, HIPSPARK_JIT_CACHE=str(cache), TZ="UTC")

    def run():
        proc = subprocess.run([sys.executable, "-c", script, str(tmp_path / "li.bin")], env=env, capture_output=True, text=True,
                              timeout=600)
        assert proc.returncode == 0, proc.stdout[-1500:] + proc.stderr[-3000:]
        return json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])

    cold = run()
    files = sorted(cache.glob("*.hsaco"))
    assert cold["hits"] == 0 and cold["compiled"] >= 1 and cold["failures"] == 0 and len(files) == cold["compiled"]
    warm = run()
    assert warm["hits"] == cold["compiled"] and warm["failures"] == 0 and warm["rows"] == cold["rows"]
    assert warm["first_query_s"] < cold["first_query_s"]
    files[0].write_bytes(files[0].read_bytes()[:-7] + b"damaged")  # checksum no longer matches: compiled again
    again = run()
    assert again["hits"] == cold["compiled"] - 1 and again["failures"] == 0 and again["rows"] == cold["rows"]
