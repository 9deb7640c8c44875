"""Expression IR of the hot path: what WHERE predicates, projections and aggregate arguments are made of.

Host-side mirror of the reference's ``mini_spark.sql`` surface (reference: src/mini_spark/sql.py) so that
queries written against the reference (``Col("a") * Col("b")``, ``F.sum(...)``, ``.alias()``, ``.like()``,
``.between()``) build the same trees with the same *auto-generated column names* (they are visible in
result rows: ``quantity_add_lit_3``, ``sum_quantity``, ``count`` ... sql.py:260,369,409,464).

Nothing in here evaluates rows.  Trees are (1) type-checked with the reference's promotion rules
(``infer_type``; sql.py:277-303) and (2) lowered to device bytecode by :mod:`minispark_amd.lowering`,
which dispatches on class *names* and the attribute names kept here (``left_side``, ``right_side``,
``operator``, ``original_col``, ``pattern``, ``value``, ``type``) - so the reference's own objects lower
through the same code when the engine is plugged into the reference.
"""

from __future__ import annotations

import operator as _op
import re
from datetime import datetime
from typing import Any, Callable, Iterable, Iterator

from .constants import ColumnType, ColumnTypePython, Schema

BINOP_SYMBOLS: dict[Callable[..., Any], str] = {
    _op.add: "+",
    _op.sub: "-",
    _op.mul: "*",
    _op.truediv: "/",
    _op.floordiv: "//",
    _op.mod: "%",
    _op.eq: "==",
    _op.ne: "!=",
    _op.lt: "<",
    _op.le: "<=",
    _op.gt: ">",
    _op.ge: ">=",
    _op.and_: "and",
    _op.or_: "or",
}


def _wrap(value: Any) -> "Col":
    return value if isinstance(value, Col) else Lit(value)


class Col:
    """Reference to a column by name; also the base class of every expression node."""

    def __init__(self, name: str) -> None:
        self.name = name

    # -- expression builders ---------------------------------------------------------------------------
    def _bin(self, other: Any, fn: Callable[..., Any]) -> "Col":
        return BinaryOperatorColumn(self, _wrap(other), fn)

    def __add__(self, o: Any) -> "Col":
        return self._bin(o, _op.add)

    def __sub__(self, o: Any) -> "Col":
        return self._bin(o, _op.sub)

    def __mul__(self, o: Any) -> "Col":
        return self._bin(o, _op.mul)

    def __truediv__(self, o: Any) -> "Col":
        return self._bin(o, _op.truediv)

    def __floordiv__(self, o: Any) -> "Col":
        return self._bin(o, _op.floordiv)

    def __mod__(self, o: Any) -> "Col":
        return self._bin(o, _op.mod)

    def __lt__(self, o: Any) -> "Col":
        return self._bin(o, _op.lt)

    def __le__(self, o: Any) -> "Col":
        return self._bin(o, _op.le)

    def __gt__(self, o: Any) -> "Col":
        return self._bin(o, _op.gt)

    def __ge__(self, o: Any) -> "Col":
        return self._bin(o, _op.ge)

    def __eq__(self, o: Any) -> "Col":  # type: ignore[override]
        return self._bin(o, _op.eq)

    def __ne__(self, o: Any) -> "Col":  # type: ignore[override]
        return self._bin(o, _op.ne)

    def __and__(self, o: Any) -> "Col":
        return self._bin(o, _op.and_)

    def __or__(self, o: Any) -> "Col":
        return self._bin(o, _op.or_)

    def __invert__(self) -> "Col":
        raise NotImplementedError  # same as the reference (sql.py:44-45)

    def __hash__(self) -> int:
        return hash((type(self).__name__, self.name))

    def like(self, pattern: str) -> "Col":
        return LikeColumn(self, pattern)

    def between(self, start: Any, end: Any) -> "Col":
        # (start <= self) & (self <= end), written so that a plain-string bound works too
        return (self >= start) & (self <= end)

    def alias(self, name: str) -> "Col":
        return AliasColumn(self, name)

    # -- tree protocol ---------------------------------------------------------------------------------
    @property
    def children(self) -> tuple["Col", ...]:
        return ()

    @property
    def all_nested_columns(self) -> Iterator["Col"]:
        yield self
        for child in self.children:
            yield from child.all_nested_columns

    def normalize_agg_columns(self) -> "Col":
        return self

    def infer_type(self, schema: Schema) -> ColumnType:
        for col_name, col_type in schema:
            if col_name == self.name:
                return col_type
        raise ValueError(f'Column "{self.name}" not found in schema {schema}')

    def __str__(self) -> str:
        return self.name

    __repr__ = __str__


class Lit(Col):
    def __init__(self, value: ColumnTypePython) -> None:
        self.value = value
        super().__init__(f"lit_{value}")

    def __hash__(self) -> int:
        return hash(("Lit", self.value))

    @property
    def all_nested_columns(self) -> Iterator[Col]:
        return iter(())

    def infer_type(self, schema: Schema) -> ColumnType:
        return ColumnType.of(self.value)

    def __str__(self) -> str:
        return str(self.value)

    __repr__ = __str__


class AliasColumn(Col):
    def __init__(self, original_col: Col, name: str) -> None:
        self.original_col = original_col
        super().__init__(name)

    def __hash__(self) -> int:
        return hash(("AliasColumn", hash(self.original_col), self.name))

    @property
    def children(self) -> tuple[Col, ...]:
        return (self.original_col,)

    def infer_type(self, schema: Schema) -> ColumnType:
        return self.original_col.infer_type(schema)

    def __str__(self) -> str:
        return f"({self.original_col}) AS {self.name}"

    __repr__ = __str__


class LikeColumn(Col):
    """SQL LIKE: ``%`` = any run of characters, ``_`` = exactly one; anchored, case-sensitive.

    ``regex`` is kept for information (reference sql.py:178-179); the device matcher works on the
    pattern itself (no regex engine on the GPU)."""

    def __init__(self, original_col: Col, pattern: str) -> None:
        self.original_col = original_col
        self.pattern = pattern
        self.regex = "^" + re.escape(pattern).replace("%", ".*").replace("_", ".") + "$"
        super().__init__(f"{original_col.name}_like_{pattern}")

    def __hash__(self) -> int:
        return hash(("LikeColumn", hash(self.original_col), self.pattern))

    @property
    def children(self) -> tuple[Col, ...]:
        return (self.original_col,)

    def infer_type(self, schema: Schema) -> ColumnType:
        if self.original_col.infer_type(schema) != ColumnType.STRING:
            raise AssertionError("LIKE operator can only be applied to string columns")
        return ColumnType.STRING  # the reference has no BOOL type (sql.py:202-205)

    def __str__(self) -> str:
        return f"({self.original_col}) LIKE '{self.pattern}'"

    __repr__ = __str__


class BinaryOperatorColumn(Col):
    def __init__(self, left_side: Any, right_side: Any, operator: Callable[..., Any]) -> None:
        self.left_side = _wrap(left_side)
        self.right_side = _wrap(right_side)
        self.operator = operator
        self.left_type_convert_to: ColumnType | None = None
        self.right_type_convert_to: ColumnType | None = None
        super().__init__(f"{self.left_side.name}_{operator.__name__}_{self.right_side.name}")

    def __hash__(self) -> int:
        return hash(("BinaryOperatorColumn", hash(self.left_side), hash(self.right_side), self.operator))

    @property
    def children(self) -> tuple[Col, ...]:
        return (self.left_side, self.right_side)

    def infer_type(self, schema: Schema) -> ColumnType:
        """Promotion rules of the reference (sql.py:277-303): ``/`` is always FLOAT, INT op FLOAT is
        FLOAT, a string literal next to a TIMESTAMP is parsed as an ISO date (the literal node is
        rewritten in place), anything else must have equal operand types.  A comparison's type is its
        operands' type - there is no BOOL."""
        lt = self.left_side.infer_type(schema)
        rt = self.right_side.infer_type(schema)
        if self.operator is _op.truediv:
            self.left_type_convert_to = None if lt == ColumnType.FLOAT else ColumnType.FLOAT
            self.right_type_convert_to = None if rt == ColumnType.FLOAT else ColumnType.FLOAT
            return ColumnType.FLOAT
        if {lt, rt} == {ColumnType.INTEGER, ColumnType.FLOAT}:
            self.left_type_convert_to = ColumnType.FLOAT if lt == ColumnType.INTEGER else None
            self.right_type_convert_to = ColumnType.FLOAT if rt == ColumnType.INTEGER else None
            return ColumnType.FLOAT
        if lt == ColumnType.STRING and rt == ColumnType.TIMESTAMP:
            lt = self._literal_to_timestamp(self.left_side)
        if rt == ColumnType.STRING and lt == ColumnType.TIMESTAMP:
            rt = self._literal_to_timestamp(self.right_side)
        if lt != rt:
            raise TypeError(f"Type mismatch in binary operation: {lt} {self.operator} {rt}")
        return lt

    @staticmethod
    def _literal_to_timestamp(side: Col) -> ColumnType:
        if type(side).__name__ != "Lit":
            raise AssertionError("only a literal can be converted to TIMESTAMP")
        side.value = datetime.fromisoformat(str(side.value))  # type: ignore[attr-defined]
        return ColumnType.TIMESTAMP

    def normalize_agg_columns(self) -> Col:
        return BinaryOperatorColumn(
            self.left_side.normalize_agg_columns(), self.right_side.normalize_agg_columns(), self.operator
        )

    def extract_left_right_key(self, left_schema: Schema, right_schema: Schema) -> tuple[Col, Col]:
        """Which side of an equi-join condition belongs to which input (reference sql.py:343-355)."""
        a, b = self.left_side, self.right_side
        if type(a) is not Col or type(b) is not Col:
            raise AssertionError("join keys must be plain columns")
        if a.name == b.name:
            raise AssertionError("Join keys must be different columns")
        left_names = {n for n, _ in left_schema}
        right_names = {n for n, _ in right_schema}
        if a.name in left_names and b.name in right_names:
            return a, b
        if a.name in right_names and b.name in left_names:
            return b, a
        raise ValueError("Join keys must be from different tables")

    def __str__(self) -> str:
        return f"({self.left_side}) {BINOP_SYMBOLS[self.operator]} ({self.right_side})"

    __repr__ = __str__


class AggCol(Col):
    """``type`` in {"sum","min","max","avg"} applied to ``original_col`` (reference sql.py:399-446)."""

    def __init__(self, agg_type: str, original_col: Col) -> None:
        self.original_col = original_col
        self.type = agg_type
        super().__init__(f"{agg_type}_{original_col.name}")

    def __hash__(self) -> int:
        return hash(("AggCol", self.type, hash(self.original_col), self.name))

    def alias(self, name: str) -> "AggCol":  # renames in place and stays an AggCol (sql.py:421-423)
        self.name = name
        return self

    @property
    def children(self) -> tuple[Col, ...]:
        return (self.original_col,)

    def infer_type(self, schema: Schema) -> ColumnType:
        return ColumnType.FLOAT if self.type == "avg" else self.original_col.infer_type(schema)

    def normalize_agg_columns(self) -> Col:
        return Col(self.name)

    def expand_avg(self) -> Iterable["AggCol"]:
        """AVG(x) is carried through the shuffle as SUM(x) and SUM(1) (sql.py:436-441)."""
        if self.type != "avg":
            return [self]
        return [
            AggCol("sum", self.original_col).alias(f"{self.name}_sum"),
            AggCol("sum", Lit(1)).alias(f"{self.name}_count"),
        ]

    def projection(self) -> Col:
        if self.type == "avg":
            return (Col(f"{self.name}_sum") / Col(f"{self.name}_count")).alias(self.name)
        return Col(self.name)

    def __str__(self) -> str:
        return f"{self.type}({self.original_col}) AS {self.name}"

    __repr__ = __str__


class Functions:
    @staticmethod
    def min(col: Col) -> AggCol:
        return AggCol("min", col)

    @staticmethod
    def max(col: Col) -> AggCol:
        return AggCol("max", col)

    @staticmethod
    def sum(col: Col) -> AggCol:
        return AggCol("sum", col)

    @staticmethod
    def avg(col: Col) -> AggCol:
        return AggCol("avg", col)

    @staticmethod
    def count() -> AggCol:
        return AggCol("sum", Lit(1)).alias("count")
