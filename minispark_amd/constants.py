"""Shared vocabulary of the hot path: column type codes, block / partition sizes.

These are *file-format* and *result-semantics* constants of the reference and therefore must
match it exactly (reference: src/mini_spark/constants.py:7-23, 81-83):

* ``ROWS_PER_BLOCK``      - rows per BlockFile block = unit of partial aggregation.
* ``SHUFFLE_PARTITIONS``  - ``hash(key) % 10`` decides which join partition a row lands in; the
  partition is the unit of partial aggregation after a join, so it is part of the result
  semantics, not a tuning knob.
* ``MAX_INT`` / ``MIN_INT`` - identities of MIN / MAX aggregates (also for FLOAT columns).
* ``ColumnType`` ordinals - the ``u8 type`` byte in the BlockFile schema header.
"""

from __future__ import annotations

import enum
from datetime import datetime
from pathlib import Path
from typing import Union

ROWS_PER_BLOCK = 2 * 1024 * 1024
SHUFFLE_PARTITIONS = 10
GLOBAL_TEMP_FOLDER = Path("tmp/")
SHUFFLE_FOLDER = Path("shuffle/")

MAX_INT = 2**31 - 1
MIN_INT = -(2**31)


class ColumnType(enum.Enum):
    """Logical column type. ``ordinal`` is the on-disk type byte, ``type`` the Python row type."""

    INTEGER = 0
    STRING = 1
    FLOAT = 2
    TIMESTAMP = 3
    UNKNOWN = 255

    @property
    def ordinal(self) -> int:
        return int(self.value)

    @property
    def type(self) -> type:
        return _PY_TYPES[self]

    @staticmethod
    def from_ordinal(ordinal: int) -> "ColumnType":
        try:
            return ColumnType(ordinal)
        except ValueError:
            raise NotImplementedError(ordinal) from None

    @staticmethod
    def of(value: object) -> "ColumnType":
        # exact type match on purpose: bool is not INTEGER (same rule as the reference)
        return _BY_PY_TYPE.get(type(value), ColumnType.UNKNOWN)

    def __str__(self) -> str:
        return self.name

    __repr__ = __str__


_PY_TYPES = {
    ColumnType.INTEGER: int,
    ColumnType.STRING: str,
    ColumnType.FLOAT: float,
    ColumnType.TIMESTAMP: int,
    ColumnType.UNKNOWN: type(None),
}
_BY_PY_TYPE = {
    int: ColumnType.INTEGER,
    str: ColumnType.STRING,
    float: ColumnType.FLOAT,
    datetime: ColumnType.TIMESTAMP,
}

ColumnTypePython = Union[int, float, str, datetime]
Row = dict
Columns = tuple
Schema = list
