"""Shared test plumbing: markers, TZ pin (BlockFile timestamps are naive local datetimes - the fixtures
were made under TZ=UTC), scratch folders, golden-fixture helpers."""

from __future__ import annotations

import json
import os
import sys
import time
from datetime import datetime
from pathlib import Path

import pytest

os.environ["TZ"] = "UTC"
time.tzset()

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def scratch_folders(tmp_path, monkeypatch):
    """Engines write result files under constants.SHUFFLE_FOLDER: point it into tmp_path."""
    from minispark_amd import constants

    monkeypatch.setattr(constants, "SHUFFLE_FOLDER", tmp_path / "shuffle")
    monkeypatch.setattr(constants, "GLOBAL_TEMP_FOLDER", tmp_path / "tmp")


from oracle.compare import assert_rows_match, f32, f32_ulps, sort_rows  # noqa: E402,F401  (re-exported)


def decode_golden_value(v):
    if isinstance(v, dict) and "f" in v:
        return float.fromhex(v["f"])
    if isinstance(v, dict) and "t" in v:
        return datetime.fromisoformat(v["t"])
    return v


def load_golden(name: str) -> dict:
    data = json.loads((GOLDEN / f"{name}.json").read_text())
    if "rows" in data:
        data["rows"] = [{k: decode_golden_value(v) for k, v in row.items()} for row in data["rows"]]
    data["paths"] = {t: str(GOLDEN / f) for t, f in data["tables"].items()}
    return data
