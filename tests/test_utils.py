"""csv -> BlockFile ingest (minispark_amd/utils.py) against the file the reference's own writer produced for
the same csv (tests/golden/ingest.bin, made by tests/golden/make_csv_golden.py)."""

from __future__ import annotations

import pytest

from minispark_amd import constants
from minispark_amd.constants import ColumnType as T
from minispark_amd.io import BlockFile
from minispark_amd.utils import convert_csv_to_block_file
from tests.conftest import GOLDEN

SCHEMA = [("id", T.INTEGER), ("price", T.FLOAT), ("name", T.STRING), ("day", T.TIMESTAMP)]


@pytest.mark.parametrize("batch", [1, 5, 8, 1000])
def test_converted_file_is_byte_identical_to_the_reference_writers(tmp_path, monkeypatch, batch):
    monkeypatch.setattr(constants, "ROWS_PER_BLOCK", 8)  # the fixture was written with 8-row blocks
    out = tmp_path / "ingest.bin"
    convert_csv_to_block_file(GOLDEN / "ingest.csv", out, SCHEMA, batch_size=batch)
    assert out.read_bytes() == (GOLDEN / "ingest.bin").read_bytes()
    rows = list(BlockFile(out).read_data_rows())
    assert len(rows) == 23 and rows[1]["name"] == "with, comma" and rows[2]["name"] == 'quote "inside"'


def test_header_only_csv_and_refusals(tmp_path):
    out = tmp_path / "empty.bin"
    convert_csv_to_block_file(GOLDEN / "ingest_empty.csv", out, SCHEMA)
    assert out.read_bytes() == (GOLDEN / "ingest_empty.bin").read_bytes()
    with pytest.raises(FileExistsError):
        convert_csv_to_block_file(GOLDEN / "ingest_empty.csv", out, SCHEMA)
    bad = tmp_path / "bad.csv"
    bad.write_text("id,price,name,day\n3000000000,1.0,x,2024-01-01\n")
    with pytest.raises(OverflowError):
        convert_csv_to_block_file(bad, tmp_path / "bad.bin", SCHEMA)
    bad.write_text("id,price,name,day\n1,1e39,x,2024-01-01\n")
    with pytest.raises(OverflowError):
        convert_csv_to_block_file(bad, tmp_path / "bad2.bin", SCHEMA)
    bad.write_text("id,price,name,day\n1,1.0,x\n")
    with pytest.raises(ValueError):
        convert_csv_to_block_file(bad, tmp_path / "bad3.bin", SCHEMA)


def test_tracer_writes_chrome_trace_json(tmp_path):
    import json

    from minispark_amd.tracing import Tracer

    t = Tracer()
    gpu = t.new_track("GPU 0")
    t.start("query")
    t.start("stage 0")
    t.end()
    t.complete("scan kernel", 1_000_000, 250_000, gpu, rows=42)
    t.end()
    t.save(tmp_path / "trace.json")
    data = json.loads((tmp_path / "trace.json").read_text())
    slices = [e for e in data["traceEvents"] if e["ph"] == "X"]
    assert [e["name"] for e in slices] == ["stage 0", "scan kernel", "query"]
    assert slices[1]["tid"] == gpu and slices[1]["dur"] == 250.0 and slices[1]["args"] == {"rows": 42}
    assert {e["args"]["name"] for e in data["traceEvents"] if e["ph"] == "M"} == {"Main System", "GPU 0"}


def test_committed_pmc_traffic_belongs_to_the_current_scan_kernel_sources():
    """bench.py quotes the committed PMC file of a config (bench.TRAFFIC_KERNELS) only while the scan kernel's sources hash to the
    measured ones (otherwise `roofline.traffic` is null).  Stale after a kernel change is a reminder to re-run the two
    --pmc passes (tools/pmc_traffic.py), not an error: the test then skips with that message."""
    import pytest

    import bench

    for config in bench.TRAFFIC_KERNELS:
        assert set(bench.pmc_traffic(config)) >= {"traffic", "traffic_source"}
    got = bench.pmc_traffic("q1")
    if got["traffic"] is None:
        pytest.skip(got["traffic_source"])
    assert 0.9 * 26 * 600_037_902 < got["traffic"] < 1.1 * 26 * 600_037_902
