"""Golden for the csv -> BlockFile ingest utility, build container only:

    TZ=UTC python tests/golden/make_csv_golden.py

The reference's converter lives in mini_spark/utils.py, which Python 3.10 cannot import (PEP 695 syntax), so the
rows of tests/golden/ingest.csv are fed to the reference's REAL BlockFile writer (mini_spark.io, imported through
the same shim as make_golden.py) in the batches its converter forms (utils.py:179-203: batch_size + 1 rows per
append, append-merge rule of io.py:231-252), with a small ROWS_PER_BLOCK so the merge rule is exercised.  Output:
tests/golden/ingest.bin (+ ingest_empty.bin for a header-only csv) - data fixtures."""

from __future__ import annotations

import csv
import sys
from datetime import datetime
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from make_golden import import_reference  # noqa: E402

ROWS_PER_BLOCK = 8
BATCH = 5


def main() -> None:
    _df, _ex, io, _sql, _tasks = import_reference()
    import mini_spark.constants as constants

    T = constants.ColumnType
    io.ROWS_PER_BLOCK = ROWS_PER_BLOCK  # the knob the reference's own tests patch (tests/test_io.py:80)
    schema = [("id", T.INTEGER), ("price", T.FLOAT), ("name", T.STRING), ("day", T.TIMESTAMP)]
    for csv_name, out_name in (("ingest.csv", "ingest.bin"), ("ingest_empty.csv", "ingest_empty.bin")):
        out = HERE / out_name
        out.unlink(missing_ok=True)
        with (HERE / csv_name).open() as f:
            reader = csv.reader(f)
            next(reader)
            rows = []
            for raw in reader:
                rows.append(tuple(datetime.fromisoformat(v) if t == T.TIMESTAMP else t.type(v)
                                  for v, (_, t) in zip(raw, schema)))
        step = BATCH + 1
        batches = [rows[i: i + step] for i in range(0, len(rows), step)] or [[]]
        if rows and len(rows) % step == 0:
            batches.append([])  # the converter's final (empty) append after a full last batch
        for batch in batches:
            io.BlockFile(out, schema).append_tuples(batch)
        print(out.name, out.stat().st_size if out.exists() else "not written", "bytes,", len(rows), "rows")


if __name__ == "__main__":
    main()
