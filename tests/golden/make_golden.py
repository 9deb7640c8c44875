"""Generate the golden fixtures by running the catalogue in tests/queries.py on the REAL reference
(PythonExecutionEngine) - build container only; /root/reference does not exist on the GPU box.

    TZ=UTC python tests/golden/make_golden.py

Writes, per case: the input BlockFiles exactly as the reference's own writer produced them
(tests/golden/<case>.<table>.bin, data fixtures) and tests/golden/<case>.json holding the expected
result rows (floats as hex, datetimes as ISO strings), the reference's stage plan summary, or the
exception type the reference raised.  No reference source is copied: the reference is imported from
where it lies through an in-memory shim (typing.Self for Python 3.10, a stand-in for the tracing
module whose dependency `perfetto` is absent - the reference's own tests replace the tracer with a
mock as well, tests/conftest.py:23-28).
"""

from __future__ import annotations

import json
import os
import sys
import tempfile
import time
import types
import typing
from datetime import datetime
from pathlib import Path

os.environ["TZ"] = "UTC"
time.tzset()

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
REFERENCE_SRC = Path("/root/reference/src")


def import_reference():
    import typing_extensions

    typing.Self = typing_extensions.Self  # reference targets Python >= 3.13
    sys.dont_write_bytecode = True
    sys.path.insert(0, str(REFERENCE_SRC))
    utils = types.ModuleType("mini_spark.utils")

    class _Tracer:
        def __getattr__(self, name):
            return lambda *a, **k: 0

    utils.TRACER = _Tracer()
    utils.trace = lambda name: (lambda f: f)
    utils.trace_yield = lambda name: (lambda f: f)
    utils.nice_schema = lambda schema: "" if schema is None else "[" + ", ".join(f"{n}:{t}" for n, t in schema) + "]"

    def convert_columns_to_rows(cols, schema):
        for row in zip(*cols):
            yield {n: v for v, (n, _) in zip(row, schema)}

    utils.convert_columns_to_rows = convert_columns_to_rows
    sys.modules["mini_spark.utils"] = utils
    import mini_spark.dataframe as df
    import mini_spark.execution as ex
    import mini_spark.io as io
    import mini_spark.sql as sql
    import mini_spark.tasks as tasks

    return df, ex, io, sql, tasks


def encode_value(v):
    if type(v) is float:
        return {"f": v.hex()}
    if type(v) is datetime:
        return {"t": v.isoformat()}
    return v


def main() -> None:
    sys.path.insert(0, str(ROOT))
    from tests.queries import CASES, api_namespace

    df, ex, io, sql, tasks = import_reference()
    only = set(sys.argv[1:])
    for case in CASES:
        if only and case.name not in only:
            continue
        # the reference writes shuffle/ relative to cwd (constants.py:11); a fresh directory per case,
        # like its tests' tmp_path (tests/conftest.py:12-20) - a failed query leaves files behind
        os.chdir(tempfile.mkdtemp(prefix=f"golden_{case.name}_"))
        paths = {}
        for spec in case.tables:
            dst = HERE / f"{case.name}.{spec.name}.bin"
            io.ROWS_PER_BLOCK = spec.rows_per_block  # same knob as reference tests/test_io.py:80
            io.BlockFile(dst).write_rows(spec.rows())
            paths[spec.name] = str(dst)
        io.ROWS_PER_BLOCK = case.rows_per_block

        class EngineDF(df.DataFrame):
            pass

        out = {"case": case.name, "tables": {s.name: f"{case.name}.{s.name}.bin" for s in case.tables}}
        with ex.PythonExecutionEngine() as engine:
            api = api_namespace(lambda: df.DataFrame(engine), sql.Col, sql.Functions, sql.Lit)
            frame = case.build(api, paths)
            try:
                plan = engine.generate_physical_plan(__import__("copy").deepcopy(frame.task))
                out["stages"] = [str(s) for s in plan.stages]
                out["schema"] = [[n, str(t)] for n, t in plan.stages[-1].writer.inferred_schema]
                rows = frame.collect()
                out["rows"] = [{k: encode_value(v) for k, v in row.items()} for row in rows]
            except Exception as e:  # noqa: BLE001 - the exception type IS the expected result
                if case.expect_error is None:
                    raise
                out["error"] = type(e).__name__
        if case.expect_error is not None and out.get("error") != case.expect_error:
            raise SystemExit(f"{case.name}: expected {case.expect_error}, reference gave {out.get('error', 'rows')}")
        (HERE / f"{case.name}.json").write_text(json.dumps(out, indent=1) + "\n")
        print(f"{case.name}: {len(out.get('rows', []))} rows {out.get('error', '')}")


if __name__ == "__main__":
    main()
