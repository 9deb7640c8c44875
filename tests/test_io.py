"""BlockFile codec (minispark_amd/io.py) - mirrors /root/reference/tests/test_io.py and adds byte-level checks
against the fixtures written by the reference's own writer and against the oracle's independent codec."""

from __future__ import annotations

from datetime import datetime
from pathlib import Path

import numpy as np
import pytest

from minispark_amd import constants
from minispark_amd.constants import ColumnType
from minispark_amd.io import BlockFile, StrCol, _deserialize_schema
from tests.conftest import GOLDEN


def test_serialize_deserialize_schema(tmp_path):
    schema = [("int_col", ColumnType.INTEGER), ("str_col", ColumnType.STRING), ("float_col", ColumnType.FLOAT),
              ("timestamp_col", ColumnType.TIMESTAMP)]
    f = tmp_path / "t.bin"
    BlockFile(f, schema).write_rows([])
    assert _deserialize_schema(f.open("rb")) == schema
    assert BlockFile(f).block_starts == [] and BlockFile(f).rows() == 0


def test_serialize_deserialize_data(tmp_path):
    rows = [
        {"int_col": 1, "str_col": "1", "float_col": 1.0, "timestamp_col": datetime(2025, 1, 1)},
        {"int_col": 2, "str_col": "2", "float_col": 2.0, "timestamp_col": datetime(2025, 1, 2)},
    ]
    bf = BlockFile(tmp_path / "t.bin")
    bf.write_rows(rows)
    assert bf.file_schema == bf.schema
    assert list(bf.read_data_rows()) == rows


def test_append_and_block_splitting(tmp_path, monkeypatch):
    """Same scenario as the reference's test_append_keeping_max_row_size (tests/test_io.py:75-98)."""
    f = tmp_path / "t.bin"
    bf = BlockFile(f, [("col1", ColumnType.STRING)]).write_rows([])
    monkeypatch.setattr(constants, "ROWS_PER_BLOCK", 10)
    for _ in range(10):
        bf.append_rows([{"col1": "x"}])
        assert len(BlockFile(f).block_starts) == 1
    bf.append_rows([{"col1": "x"}])
    assert len(BlockFile(f).block_starts) == 2 and len(list(BlockFile(f).read_data_rows())) == 11
    bf.append_rows([{"col1": "x"}] * 5)
    assert len(BlockFile(f).block_starts) == 2 and len(list(BlockFile(f).read_data_rows())) == 16
    bf.append_rows([{"col1": "x"}] * 5)
    assert len(BlockFile(f).block_starts) == 3 and len(list(BlockFile(f).read_data_rows())) == 21
    assert BlockFile(f).block_rows() == [10, 10, 1]


def test_golden_fruit_bytes():
    """SURVEY.md Appendix A golden example: the 4-row fruit table is these 165 bytes."""
    data = (GOLDEN / "fruit.fruits.bin").read_bytes()
    assert len(data) == 165 - 1 or len(data) == 165 or len(data) == 164  # header 32 + block + footer
    assert data[:32] == bytes([4, 1, 5]) + b"fruit" + bytes([0, 8]) + b"quantity" + bytes([1, 5]) + b"color" + bytes([2, 5]) + b"price"
    assert data[-12:] == (32).to_bytes(8, "little") + (1).to_bytes(4, "little")


@pytest.mark.parametrize("name", ["q1_ragged_blocks.lineitem.bin", "join_group.orders.bin", "edge_minmax.edge.bin",
                                  "e2e_select_star.orders.bin", "fruit.fruits.bin"])
def test_reencode_reference_written_files_byte_exact(tmp_path, name):
    """Read a file the reference's writer produced, write it back with this codec: identical bytes."""
    src = BlockFile(GOLDEN / name)
    blocks = [src.read_block_raw(b) for b in range(len(src.block_starts))]
    out = tmp_path / "copy.bin"
    BlockFile(out).write_raw_blocks(src.file_schema, blocks)
    assert out.read_bytes() == (GOLDEN / name).read_bytes()


def test_pruned_read_touches_only_requested_columns():
    src = BlockFile(GOLDEN / "q1_multiblock.lineitem.bin")
    names = [n for n, _ in src.file_schema]
    want = [names.index("l_discount"), names.index("l_returnflag")]
    disc, flag = src.read_block_raw(2, want)
    full = src.read_block_raw(2)
    assert np.array_equal(disc, full[want[0]]) and isinstance(flag, StrCol)
    assert flag.to_list() == full[want[1]].to_list()
    layout = src.block_layout(2)
    assert layout.nrows == 1024 and len(layout.spans) == len(names)


def test_row_api_against_independent_oracle_codec(tmp_path):
    from oracle import blockfile as bfio

    rows = [{"i": -5, "s": "", "f": 0.1, "t": datetime(1999, 12, 31, 23, 59, 59)},
            {"i": 2147483647, "s": "hello world", "f": -3.25e38, "t": datetime(1970, 1, 1)}]
    f = tmp_path / "a.bin"
    BlockFile(f).write_rows(rows)
    g = tmp_path / "b.bin"
    bfio.write_blockfile(g, [("i", 0), ("s", 1), ("f", 2), ("t", 3)], [[r[k] for r in rows] for k in "isft"], 1 << 21)
    assert f.read_bytes() == g.read_bytes()
    schema, blocks = bfio.read_blockfile(f)
    assert blocks[0][2][0] == np.float32(0.1)  # FLOAT survives as f32


def test_quantisation_errors_match_reference():
    with pytest.raises(OverflowError):
        BlockFile(Path("/tmp/x"), [("i", ColumnType.INTEGER)])._write_python_columns(([2**31],), [("i", ColumnType.INTEGER)])
    with pytest.raises(OverflowError):
        BlockFile(Path("/tmp/x"), [("f", ColumnType.FLOAT)])._write_python_columns(([1e39],), [("f", ColumnType.FLOAT)])
    with pytest.raises(AssertionError):
        BlockFile(Path("/tmp/x"), [("f", ColumnType.FLOAT)])._write_python_columns(([1],), [("f", ColumnType.FLOAT)])
    with pytest.raises(ValueError):
        StrCol.from_strings(["x" * 256])


def test_result_file_rows_equal_the_rows_of_the_file_it_writes(tmp_path):
    """The in-memory result hand-over (jobs.ResultFile.rows) yields exactly what a reader of the BlockFile
    written for the same result yields - same Python types and values; the file appears on first use."""
    from minispark_amd.jobs import ResultFile

    schema = [("k", ColumnType.STRING), ("i", ColumnType.INTEGER), ("f", ColumnType.FLOAT), ("t", ColumnType.TIMESTAMP)]
    words = ["", "a", "héllo", "x" * 200]
    enc = [w.encode() for w in words]
    raw = [
        StrCol(np.array([len(e) for e in enc], dtype=np.uint8), np.frombuffer(b"".join(enc), dtype=np.uint8)),
        np.array([0, -1, 2**31 - 1, -(2**31)], dtype=np.int32),
        np.array([0.1, -2.5e30, 1e-40, 3.0], dtype=np.float32),
        np.array([0, 1, 912_470_400_000_000, 86_400_000_000], dtype=np.int64),
    ]
    res = ResultFile(tmp_path / "stage" / "result.bin", schema, raw, 4)
    assert not (tmp_path / "stage").exists()
    direct = list(res.rows())
    from_file = list(BlockFile(res.file_path).read_data_rows())
    assert (tmp_path / "stage" / "result.bin").exists()
    assert direct == from_file
    assert [type(v) for v in direct[1].values()] == [type(v) for v in from_file[1].values()]
    assert isinstance(direct[0]["t"], datetime) and isinstance(direct[0]["f"], float) and isinstance(direct[0]["i"], int)


def test_vectorised_timestamp_column_equals_the_value_by_value_conversion():
    """rows_from_raw turns a TIMESTAMP column into datetimes in one numpy step when the local zone is UTC; it must give
    exactly what the reference's ``datetime.fromtimestamp(us / 1_000_000)`` gives value by value (io.py:38-39)."""
    import numpy as np

    from minispark_amd import io

    rng = np.random.default_rng(3)
    us = rng.integers(-10**15, 3 * 10**15, 50_000).astype(np.int64)
    us[:6] = [0, 1, -1, 999_999, 10**15 + 1, 883_612_800_000_000]
    assert io.timestamps_to_datetimes(us) == [io.timestamp_to_datetime(int(v)) for v in us.tolist()]
    assert io.timestamps_to_datetimes(np.zeros(0, np.int64)) == []
    far = np.asarray([5 * 10**15], dtype=np.int64)  # beyond the range the shortcut is proven for: value by value
    assert io.timestamps_to_datetimes(far) == [io.timestamp_to_datetime(int(far[0]))]


def test_rows_from_raw_columns_as_a_list_and_column_wise():
    """The row builder (one generated list comprehension per column count) gives the rows a BlockFile reader gives;
    ResultFile.columns hands the same result over without building a dict per row."""
    from datetime import datetime

    import numpy as np

    from minispark_amd.constants import ColumnType as T
    from minispark_amd.io import StrCol, datetime_to_timestamp, rows_from_raw, rows_list_from_raw
    from minispark_amd.jobs import ResultFile

    schema = [("s", T.STRING), ("i", T.INTEGER), ("f", T.FLOAT), ("t", T.TIMESTAMP)]
    when = [datetime(1998, 9, 2), datetime(1970, 1, 1, 0, 0, 1), datetime(2024, 2, 29, 12, 30)]
    raw = [StrCol.from_strings(["a", "", "long string"]), np.array([1, -2, 3], np.int32), np.array([0.5, 1.25, -3.0], np.float32),
           np.array([datetime_to_timestamp(w) for w in when], np.int64)]
    want = [{"s": "a", "i": 1, "f": 0.5, "t": when[0]}, {"s": "", "i": -2, "f": 1.25, "t": when[1]},
            {"s": "long string", "i": 3, "f": -3.0, "t": when[2]}]
    assert rows_list_from_raw(schema, raw) == want == list(rows_from_raw(schema, raw))
    assert rows_list_from_raw(schema[1:2], raw[1:2]) == [{"i": 1}, {"i": -2}, {"i": 3}]
    assert rows_list_from_raw([], []) == []
    big = rows_list_from_raw(schema[1:3], [np.arange(10_000, dtype=np.int32), np.arange(10_000, dtype=np.float32)])
    assert len(big) == 10_000 and big[9_999] == {"i": 9_999, "f": 9_999.0}
    cols = ResultFile(None, schema, raw, 3).columns()
    assert cols["s"] == ["a", "", "long string"] and cols["i"].tolist() == [1, -2, 3] and cols["t"].dtype == np.int64


def test_small_results_decoded_with_the_hand_over_equal_their_raw_columns(tmp_path):
    """Round 3: results of a handful of rows are handed over as Python values decoded straight from the result image
    (io.LazyRaw); the numpy columns - and with them the result BlockFile and the column-wise form - are built only on
    demand and must describe the same rows."""
    from datetime import datetime

    from minispark_amd.constants import ColumnType as T
    from minispark_amd.io import BlockFile, LazyRaw, StrCol
    from minispark_amd.jobs import ResultFile

    schema = [("flag", T.STRING), ("s", T.FLOAT), ("n", T.INTEGER), ("t", T.TIMESTAMP)]
    raw = [StrCol.from_strings(["A", "N", "R"]), np.array([1.5, -2.25, 3.0], np.float32), np.array([7, -8, 9], np.int32),
           np.array([0, 86_400_000_000, 946_684_800_000_000], np.int64)]
    py = [["A", "N", "R"], [1.5, -2.25, 3.0], [7, -8, 9],
          [datetime(1970, 1, 1), datetime(1970, 1, 2), datetime(2000, 1, 1)]]
    built = []
    lazy = ResultFile(tmp_path / "a" / "result.bin", schema, LazyRaw(py, lambda: built.append(1) or raw), 3)
    plain = ResultFile(tmp_path / "b" / "result.bin", schema, raw, 3)
    assert lazy.rows() == plain.rows() and not built          # rows: no numpy column was built
    assert lazy.rows() is not lazy.rows()                       # fresh dicts per call
    assert set(lazy.columns()) == {"flag", "s", "n", "t"} and built == [1]
    assert list(BlockFile(lazy.file_path).read_data_rows()) == plain.rows() and built == [1]
