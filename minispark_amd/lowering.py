"""Expression trees -> device bytecode (the stack programs of include/hipspark.h).

Works on the reference's expression objects as well as on :mod:`minispark_amd.sql`'s: dispatch is by
class *name* and by the attribute names both share (``left_side``/``right_side``/``operator``,
``original_col``, ``pattern``, ``value``, ``type``), see reference src/mini_spark/sql.py.

Typing of in-flight values (reference: rows hold Python objects, sql.py:262-266):

====== =========================== ==========================================
 tag    Python value in the oracle  device cell
====== =========================== ==========================================
 ``I``  int (unbounded)             i64
 ``F``  float (fp64)                f64
 ``B``  bool                        i64 0/1
 ``T``  datetime                    i64 microseconds (compare only)
 ``S``  str                         not a cell: string ops read the column bytes
====== =========================== ==========================================
"""

from __future__ import annotations

import struct
from dataclasses import dataclass, field
from datetime import datetime
from typing import Any, Sequence

from . import hipspark as hs
from .constants import ColumnType, Schema
from .io import datetime_to_timestamp

_ARITH = {"add": (hs.OP_ADD_I, hs.OP_ADD_F), "sub": (hs.OP_SUB_I, hs.OP_SUB_F), "mul": (hs.OP_MUL_I, hs.OP_MUL_F),
          "floordiv": (hs.OP_FLOORDIV_I, hs.OP_FLOORDIV_F), "mod": (hs.OP_MOD_I, hs.OP_MOD_F)}
_CMP = {"lt": (hs.OP_LT_I, hs.OP_LT_F, 0), "le": (hs.OP_LE_I, hs.OP_LE_F, 1), "gt": (hs.OP_GT_I, hs.OP_GT_F, 2),
        "ge": (hs.OP_GE_I, hs.OP_GE_F, 3), "eq": (hs.OP_EQ_I, hs.OP_EQ_F, 4), "ne": (hs.OP_NE_I, hs.OP_NE_F, 5)}
_FLIP = {0: 2, 1: 3, 2: 0, 3: 1, 4: 4, 5: 5}  # swap operands of a comparison
_LOGIC = {"and_": hs.OP_AND, "or_": hs.OP_OR}
_STORAGE_TAG = {hs.I32: "I", hs.I64: "I", hs.F32: "F", hs.F64: "F", hs.U8: "B"}


class LoweringError(NotImplementedError):
    """The expression is outside what the device evaluator supports."""


def _cls(node: Any) -> str:
    return type(node).__name__


def unalias(node: Any) -> Any:
    while _cls(node) in ("AliasColumn",):
        node = node.original_col
    return node


def expr_key(node: Any) -> tuple:
    """Structural identity of an expression (for de-duplicating aggregate arguments)."""
    name = _cls(node)
    if name in ("AliasColumn", "AggCol"):
        return expr_key(node.original_col)
    if name in ("Col", "SchemaCol"):
        return ("col", node.name)
    if name == "Lit":
        return ("lit", type(node.value).__name__, repr(node.value))
    if name == "LikeColumn":
        return ("like", expr_key(node.original_col), node.pattern)
    if name == "BinaryOperatorColumn":
        return (node.operator.__name__, expr_key(node.left_side), expr_key(node.right_side))
    raise LoweringError(f"unsupported expression node {name}")


class NeedsDecoded(Exception):
    """A dictionary-coded column is used in a way that has no coded form (compared with another column, part of a
    concatenation whose product dictionary is too large, ...): the caller decodes batch column ``column`` and lowers
    again."""

    def __init__(self, column: int) -> None:
        super().__init__(f"batch column {column} must be decoded")
        self.column = column


def dict_mask_words(entries: Sequence[bytes], predicate: Any) -> list[int]:
    """One bit per dictionary entry (bit c of word c // 64 = predicate(entry c)) as 64-bit literal words."""
    words = [0] * max(1, (len(entries) + 63) // 64)
    for code, entry in enumerate(entries):
        if predicate(entry.decode("utf-8")):
            words[code // 64] |= 1 << (code % 64)
    return words


@dataclass
class StringParts:
    """A STRING-valued expression flattened to the parts of a concatenation (columns or literals)."""

    parts: list  # of ("col", batch_col_index) | ("lit", bytes)


@dataclass
class Program:
    ins: list[int] = field(default_factory=list)
    lits: list[int] = field(default_factory=list)
    pool: bytearray = field(default_factory=bytearray)
    columns: list[int] = field(default_factory=list)  # slot -> batch column index
    max_depth: int = 0
    code_columns: list[int] = field(default_factory=list)  # batch columns to hand over as HS_U8 (code bytes only)

    def to_struct(self) -> hs.hs_program:
        if len(self.ins) > hs.HS_MAX_INS:
            raise LoweringError(f"program has {len(self.ins)} instructions (> {hs.HS_MAX_INS})")
        if len(self.lits) > hs.HS_MAX_LIT:
            raise LoweringError(f"program has {len(self.lits)} literals (> {hs.HS_MAX_LIT})")
        if len(self.pool) > hs.HS_MAX_POOL:
            raise LoweringError(f"program has {len(self.pool)} literal bytes (> {hs.HS_MAX_POOL})")
        p = hs.hs_program()
        p.n_ins = len(self.ins)
        p.n_lit = len(self.lits)
        for i, w in enumerate(self.ins):
            p.ins[i] = w
        for i, w in enumerate(self.lits):
            p.lit[i] = w
        for i, b in enumerate(self.pool):
            p.pool[i] = b
        return p

    def to_bytes(self) -> bytes:
        """Canonical serialisation (tests compare programs lowered from different front-ends)."""
        return (
            struct.pack("<II", len(self.ins), len(self.lits))
            + b"".join(struct.pack("<Q", w) for w in self.ins)
            + b"".join(struct.pack("<Q", w) for w in self.lits)
            + bytes(self.pool)
            + struct.pack(f"<{len(self.columns)}i", *self.columns)
        )


_COL_OPS_A = {hs.OP_LD, hs.OP_STRCMP_LIT, hs.OP_LIKE, hs.OP_DICTBIT}
# comparison codes of HS_OP_STRCMP_* (0 lt, 1 le, 2 gt, 3 ge, 4 eq, 5 ne) on Python strings (ASCII: byte order)
_STR_CMP = {0: lambda a, b: a < b, 1: lambda a, b: a <= b, 2: lambda a, b: a > b, 3: lambda a, b: a >= b,
            4: lambda a, b: a == b, 5: lambda a, b: a != b}


class ProgramBuilder:
    """Lowers expressions over one input batch.  ``schema`` names/types its columns, ``kinds`` are their
    device storage kinds (hs.I32 ... hs.STR)."""

    def __init__(self, schema: Schema, kinds: Sequence[int], dicts: Sequence[Any] | None = None) -> None:
        self.schema = list(schema)
        self.kinds = list(kinds)
        # per batch column: the dictionary (tuple of bytes) of a dictionary-coded STRING column, else None
        self.dicts = list(dicts) if dicts is not None else [None] * len(self.kinds)
        self.ins: list[tuple[int, int, int, int, int]] = []  # (op, sp, a, b, c) with a/b = BATCH col index
        self.lits: list[int] = []
        self.pool = bytearray()
        self.sp = 0
        self.max_depth = 0
        self.used: list[int] = []  # batch column indices in first-use order
        self.code_reads: set[int] = set()    # coded columns read as a code byte (HS_OP_DICTBIT)
        self.string_reads: set[int] = set()  # columns read as strings (LIKE, comparisons)

    # ---- emit helpers ------------------------------------------------------------------------------
    def _emit(self, op: int, a: int = 0, b: int = 0, c: int = 0, push: int = 0) -> None:
        self.ins.append((op, self.sp, a, b, c))
        self.sp += push
        if self.sp < 0:
            raise AssertionError("stack underflow while lowering")
        self.max_depth = max(self.max_depth, self.sp)
        if self.max_depth > hs.HS_MAX_STACK:
            raise LoweringError(f"expression needs a stack deeper than {hs.HS_MAX_STACK}")

    def _lit(self, word: int) -> int:
        word &= 0xFFFFFFFFFFFFFFFF
        if word in self.lits:
            return self.lits.index(word)
        self.lits.append(word)
        return len(self.lits) - 1

    def _lit_run(self, words: Sequence[int]) -> int:
        """Index of the first of len(words) CONSECUTIVE literal slots holding ``words`` (no de-duplication)."""
        first = len(self.lits)
        self.lits.extend(w & 0xFFFFFFFFFFFFFFFF for w in words)
        return first

    def _emit_dictbit(self, idx: int, predicate: Any) -> None:
        """bit[code] of dictionary-coded column ``idx``: the predicate is evaluated once per dictionary entry here,
        the rows only test a bit (HS_OP_DICTBIT)."""
        words = dict_mask_words(self.dicts[idx], predicate)
        if len(words) > 4:
            raise NeedsDecoded(idx)
        self.code_reads.add(idx)
        self._emit(hs.OP_DICTBIT, a=idx, b=self._lit_run(words), c=len(words), push=1)

    def _pool_ref(self, data: bytes) -> int:
        off = self.pool.find(data) if data else 0
        if off < 0 or not data:
            off = len(self.pool)
            self.pool += data
        return self._lit((off << 32) | len(data))

    def use_column(self, name: str) -> int:
        for i, (col_name, _) in enumerate(self.schema):
            if col_name == name:
                if i not in self.used:
                    self.used.append(i)
                return i
        raise ValueError(f'Column "{name}" not found in schema {self.schema}')

    # ---- expression lowering -------------------------------------------------------------------------
    def lower(self, node: Any) -> str:
        """Push the value of ``node``; returns its tag.  STRING-valued nodes cannot be pushed."""
        name = _cls(node)
        if name in ("AliasColumn", "AggCol"):
            return self.lower(node.original_col)
        if name in ("Col", "SchemaCol"):
            idx = self.use_column(node.name)
            kind = self.kinds[idx]
            if kind == hs.STR:
                raise LoweringError(f"string column {node.name} used as a number")
            self._emit(hs.OP_LD, a=idx, push=1)
            ctype = self.schema[idx][1]
            if ctype == ColumnType.TIMESTAMP:
                return "T"
            return _STORAGE_TAG[kind]
        if name == "Lit":
            return self._lower_literal(node.value)
        if name == "LikeColumn":
            target = unalias(node.original_col)
            if _cls(target) not in ("Col", "SchemaCol"):
                raise LoweringError("LIKE is supported on plain string columns only")
            idx = self.use_column(target.name)
            if self.kinds[idx] != hs.STR:
                raise AssertionError("LIKE operator can only be applied to string columns")
            if self.dicts[idx] is not None:
                import re  # noqa: PLC0415

                # the reference's own reading of the pattern (sql.py:178-179, 192-194), applied to every entry
                regex = "^" + re.escape(node.pattern).replace("%", ".*").replace("_", ".") + "$"
                self._emit_dictbit(idx, lambda text: re.match(regex, text) is not None)
                return "B"
            self.string_reads.add(idx)
            self._emit(hs.OP_LIKE, a=idx, b=self._pool_ref(node.pattern.encode("utf-8")), push=1)
            return "B"
        if name == "BinaryOperatorColumn":
            return self._lower_binary(node)
        raise LoweringError(f"unsupported expression node {name}")

    def _lower_literal(self, value: Any) -> str:
        if type(value) is bool:
            self._emit(hs.OP_LIT, a=self._lit(int(value)), push=1)
            return "B"
        if type(value) is int:
            if not -(2**63) <= value < 2**63:
                raise LoweringError("integer literal outside i64")
            self._emit(hs.OP_LIT, a=self._lit(value), push=1)
            return "I"
        if type(value) is float:
            self._emit(hs.OP_LIT, a=self._lit(struct.unpack("<Q", struct.pack("<d", value))[0]), push=1)
            return "F"
        if type(value) is datetime:
            self._emit(hs.OP_LIT, a=self._lit(datetime_to_timestamp(value)), push=1)
            return "T"
        raise LoweringError(f"literal {value!r} cannot be used as a number")

    def string_tag(self, node: Any) -> bool:
        """True when ``node`` is STRING-valued in this batch."""
        name = _cls(node)
        if name in ("AliasColumn", "AggCol"):
            return self.string_tag(node.original_col)
        if name in ("Col", "SchemaCol"):
            for i, (col_name, _) in enumerate(self.schema):
                if col_name == node.name:
                    return self.kinds[i] == hs.STR
            raise ValueError(f'Column "{node.name}" not found in schema {self.schema}')
        if name == "Lit":
            return type(node.value) is str
        if name == "BinaryOperatorColumn" and node.operator.__name__ == "add":
            return self.string_tag(node.left_side) and self.string_tag(node.right_side)
        return False

    def string_parts(self, node: Any) -> StringParts:
        """Flatten a STRING-valued expression (columns, literals, '+') into concat parts."""
        name = _cls(node)
        if name in ("AliasColumn", "AggCol"):
            return self.string_parts(node.original_col)
        if name in ("Col", "SchemaCol"):
            return StringParts([("col", self.use_column(node.name))])
        if name == "Lit":
            return StringParts([("lit", str(node.value).encode("utf-8"))])
        if name == "BinaryOperatorColumn" and node.operator.__name__ == "add":
            return StringParts(self.string_parts(node.left_side).parts + self.string_parts(node.right_side).parts)
        raise LoweringError(f"unsupported string expression {node}")

    def _lower_string_compare(self, node: Any, cmp_code: int) -> str:
        left, right = unalias(node.left_side), unalias(node.right_side)
        lcol = _cls(left) in ("Col", "SchemaCol")
        rcol = _cls(right) in ("Col", "SchemaCol")
        if lcol and rcol:
            a, b = self.use_column(left.name), self.use_column(right.name)
            for idx in (a, b):
                if self.dicts[idx] is not None:  # codes of two dictionaries do not compare
                    raise NeedsDecoded(idx)
            self.string_reads.update((a, b))
            self._emit(hs.OP_STRCMP_COL, a=a, b=b, c=cmp_code, push=1)
        elif (lcol and _cls(right) == "Lit") or (rcol and _cls(left) == "Lit"):
            col, lit, code = (left, right, cmp_code) if lcol else (right, left, _FLIP[cmp_code])
            idx = self.use_column(col.name)
            text = str(lit.value)
            if self.dicts[idx] is not None:
                self._emit_dictbit(idx, lambda entry: _STR_CMP[code](entry, text))
            else:
                self.string_reads.add(idx)
                self._emit(hs.OP_STRCMP_LIT, a=idx, b=self._pool_ref(text.encode("utf-8")), c=code, push=1)
        else:
            raise LoweringError(f"string comparison of computed strings is not supported: {node}")
        return "B"

    def _to_float(self, tag: str, second: bool) -> None:
        if tag in ("I", "B"):
            self._emit(hs.OP_I2F, a=1 if second else 0)
        elif tag != "F":
            raise TypeError(f"cannot use a {tag} value as FLOAT")

    def _lower_binary(self, node: Any) -> str:
        opname = node.operator.__name__
        if opname in _CMP and (self.string_tag(node.left_side) or self.string_tag(node.right_side)):
            lt_is_ts = self._is_timestamp(node.left_side)
            rt_is_ts = self._is_timestamp(node.right_side)
            if not (lt_is_ts or rt_is_ts):
                if not (self.string_tag(node.left_side) and self.string_tag(node.right_side)):
                    raise TypeError(f"Type mismatch in binary operation: {node}")
                return self._lower_string_compare(node, _CMP[opname][2])
        lt = self._lower_operand(node.left_side, other=node.right_side)
        rt = self._lower_operand(node.right_side, other=node.left_side)
        if opname == "truediv":
            self._to_float(lt, second=True)
            self._to_float(rt, second=False)
            self._emit(hs.OP_DIV_F, push=-1)
            return "F"
        if opname in _ARITH:
            if "T" in (lt, rt) or "S" in (lt, rt):
                raise LoweringError(f"arithmetic on {lt}/{rt} values is not supported")
            if "F" in (lt, rt):
                self._to_float(lt, second=True)
                self._to_float(rt, second=False)
                self._emit(_ARITH[opname][1], push=-1)
                return "F"
            self._emit(_ARITH[opname][0], push=-1)
            return "I"
        if opname in _CMP:
            if (lt == "T") != (rt == "T"):
                raise TypeError(f"Type mismatch in binary operation: {node}")
            if "F" in (lt, rt):
                self._to_float(lt, second=True)
                self._to_float(rt, second=False)
                self._emit(_CMP[opname][1], push=-1)
            else:
                self._emit(_CMP[opname][0], push=-1)
            return "B"
        if opname in _LOGIC:
            if lt not in ("I", "B") or rt not in ("I", "B"):
                raise TypeError(f"unsupported operand type(s) for {opname}: {lt} and {rt}")
            self._emit(_LOGIC[opname], push=-1)
            return "B" if (lt, rt) == ("B", "B") else "I"
        raise LoweringError(f"unsupported operator {opname}")

    def _is_timestamp(self, node: Any) -> bool:
        node = unalias(node)
        if _cls(node) in ("Col", "SchemaCol"):
            for col_name, ctype in self.schema:
                if col_name == node.name:
                    return ctype == ColumnType.TIMESTAMP
        if _cls(node) == "Lit":
            return type(node.value) is datetime
        return False

    def _lower_operand(self, node: Any, other: Any) -> str:
        """Like lower(), but an ISO string literal next to a TIMESTAMP operand becomes a timestamp
        (reference sql.py:291-298 rewrites the literal during type inference)."""
        bare = unalias(node)
        if _cls(bare) == "Lit" and type(bare.value) is str and self._is_timestamp(other):
            return self._lower_literal(datetime.fromisoformat(bare.value))
        return self.lower(node)

    # ---- sinks -----------------------------------------------------------------------------------
    def emit_filter(self, cond: Any) -> None:
        tag = self.lower(cond)
        if tag not in ("B", "I"):
            self._to_bool(tag)
        self._emit(hs.OP_FILTER, push=-1)

    def _to_bool(self, tag: str) -> None:
        if tag == "F":  # truthiness of a float: x != 0.0
            self._emit(hs.OP_LIT, a=self._lit(0), push=1)
            self._emit(hs.OP_NE_F, push=-1)
        else:
            raise LoweringError(f"a {tag} value cannot be used as a condition")

    def emit_key(self) -> None:
        self._emit(hs.OP_KEY)

    def emit_agg(self, acc: int, arg: Any) -> str:
        tag = self.lower(arg)
        if tag in ("T", "S"):
            raise AssertionError("aggregate argument must be numeric")  # reference: tasks.py:298
        self._emit(hs.OP_AGG, a=acc, push=-1)
        return tag

    def emit_out(self, out: int, expr: Any) -> str:
        tag = self.lower(expr)
        self._emit(hs.OP_OUT, a=out, push=-1)
        return tag

    # ---- finish --------------------------------------------------------------------------------------
    def finish(self, key_column: int | None = None) -> Program:
        """Assign column slots (numeric columns and the key first, strings after) and encode."""
        if self.sp != 0:
            raise AssertionError("unbalanced program")
        if key_column is not None and key_column not in self.used:
            self.used.append(key_column)
        # a coded column that is only ever read as its code byte travels as a HS_U8 column: preloaded with the numeric
        # ones (one 4-byte load per row quad) instead of one byte load per row
        as_bytes = {i for i in self.code_reads if i not in self.string_reads and i != key_column}
        numeric = [i for i in self.used if self.kinds[i] != hs.STR or i == key_column or i in as_bytes]
        strings = [i for i in self.used if self.kinds[i] == hs.STR and i != key_column and i not in as_bytes]
        order = numeric + strings
        if len(order) > hs.HS_MAX_COLS:
            raise LoweringError(f"expression reads {len(order)} columns (> {hs.HS_MAX_COLS})")
        slot = {col: s for s, col in enumerate(order)}
        words = []
        for op, sp, a, b, c in self.ins:
            if op in _COL_OPS_A:
                a = slot[a]
            elif op == hs.OP_STRCMP_COL:
                a, b = slot[a], slot[b]
            words.append(op | (sp << 8) | (a << 16) | (b << 32) | (c << 48))
        return Program(words, list(self.lits), bytearray(self.pool), order, self.max_depth, sorted(as_bytes))


AGG_CODES = {"sum": hs.AGG_SUM, "min": hs.AGG_MIN, "max": hs.AGG_MAX}


@dataclass
class AggregateLowering:
    program: Program
    key_slot: int
    acc_ops: list[int]          # per accumulator: hs.AGG_*
    acc_is_int: list[bool]
    agg_to_acc: list[int]       # per requested aggregate column: its accumulator
    numeric_slots: int          # slots that must sit in the preloaded range

    def spec(self) -> hs.hs_agg_spec:
        s = hs.hs_agg_spec()
        s.n_acc = len(self.acc_ops)
        for i, (op, is_int) in enumerate(zip(self.acc_ops, self.acc_is_int)):
            s.op[i] = op
            s.is_int[i] = 1 if is_int else 0
        return s


def lower_aggregate(schema: Schema, kinds: Sequence[int], filters: Sequence[Any], group_by: Any,
                    agg_columns: Sequence[Any], dicts: Sequence[Any] | None = None) -> AggregateLowering:
    """[filter ... FILTER]* KEY [arg ... AGG acc]* for one partial-aggregate launch.

    Aggregates with the same function and structurally equal argument share an accumulator (Q1's
    eleven aggregate columns need six)."""
    b = ProgramBuilder(schema, kinds, dicts)
    for cond in filters:
        b.emit_filter(cond)
    key = unalias(group_by)
    if _cls(key) not in ("Col", "SchemaCol"):
        raise ValueError(f"Unknown columns in GroupBy: {[getattr(key, 'name', key)]}")
    key_col = b.use_column(key.name)
    b.emit_key()
    acc_index: dict[tuple, int] = {}
    acc_ops: list[int] = []
    acc_is_int: list[bool] = []
    agg_to_acc: list[int] = []
    for agg in agg_columns:
        if agg.type not in AGG_CODES:
            raise LoweringError(f"aggregate {agg.type} must be expanded before lowering")
        ident = (agg.type, expr_key(agg.original_col))
        if ident not in acc_index:
            acc = len(acc_ops)
            if acc >= hs.HS_MAX_ACC:
                raise LoweringError(f"more than {hs.HS_MAX_ACC} distinct aggregates")
            tag = b.emit_agg(acc, agg.original_col)
            acc_index[ident] = acc
            acc_ops.append(AGG_CODES[agg.type])
            acc_is_int.append(tag in ("I", "B"))
        agg_to_acc.append(acc_index[ident])
    prog = b.finish(key_column=key_col)
    numeric_slots = sum(1 for c in prog.columns if kinds[c] != hs.STR or c == key_col or c in prog.code_columns)
    return AggregateLowering(prog, prog.columns.index(key_col), acc_ops, acc_is_int, agg_to_acc, numeric_slots)


class FinishUnsupported(Exception):
    """The final merge + projection does not fit the one-launch tail (the caller takes the general operator sequence)."""


_FILE_KIND = {ColumnType.INTEGER: hs.I32, ColumnType.FLOAT: hs.F32, ColumnType.TIMESTAMP: hs.I64, ColumnType.STRING: hs.STR}


def lower_finish(agg_to_acc: Sequence[int], acc_kinds: Sequence[int], key_kind: int, agg_columns: Sequence[Any],
                 merged_schema: Schema, project: Sequence[Any] | None, out_schema: Schema) -> tuple:
    """The final stage - merge of the partial rows (reference tasks.py:290-292), the projection after it
    (plan.py:190-203: AVG = sum / count, renames) and the stored kinds of the result file (io.py:87-94) - as the
    description hs_agg_finish takes: -> (hs_finish_spec without offsets, hs_program or None, [(src, index, stored kind)]).
    Pure lowering (no device): used by the engine's short tail and by the stage-level ABI (minispark_amd/stage.py)."""
    ops = {"sum": hs.AGG_SUM, "min": hs.AGG_MIN, "max": hs.AGG_MAX}
    fin = hs.hs_finish_spec()
    folds: dict[tuple[int, int], int] = {}
    col_fold: list[int] = []  # merged column i + 1 -> fold
    for i, agg in enumerate(agg_columns):
        pair = (agg_to_acc[i], ops[agg.type])
        if pair not in folds:
            if len(folds) >= hs.HS_MAX_ACC:
                raise FinishUnsupported("too many aggregates for the fused tail")
            folds[pair] = len(folds)
            fin.fold_src[folds[pair]], fin.fold_op[folds[pair]] = pair
        col_fold.append(folds[pair])
    fin.n_fold = len(folds)
    merged_kinds = [key_kind] + [hs.I64 if acc_kinds[agg_to_acc[i]] == hs.I32 else hs.F64
                                 for i in range(len(agg_columns))]
    if len(merged_schema) != len(merged_kinds):
        raise AssertionError(f"merge schema {merged_schema} does not match {len(agg_columns)} aggregates")

    def stored(in_kind: int, col_type: ColumnType) -> int:
        want = _FILE_KIND[col_type]
        if (in_kind, want) in ((hs.F64, hs.F32), (hs.I64, hs.I32), (hs.I64, hs.I64)):
            return want
        raise AssertionError(f"column of kind {in_kind} cannot be stored as {col_type}")

    outs: list[tuple[int, int, int]] = []  # (src, index, stored kind)
    prog = None
    if project is None:
        if len(out_schema) != len(merged_kinds):
            raise AssertionError(f"writer schema {out_schema} does not match merged columns {merged_schema}")
        outs.append((0, 0, key_kind))
        for i in range(len(agg_columns)):
            outs.append((1, col_fold[i], stored(merged_kinds[i + 1], out_schema[i + 1][1])))
    else:
        if len(out_schema) != len(project):
            raise AssertionError(f"writer schema {out_schema} does not match the projection")
        b = ProgramBuilder(list(merged_schema), merged_kinds)
        names = [n for n, _ in merged_schema]
        n_prog = 0
        for o, col in enumerate(project):
            bare = unalias(col)
            if _cls(bare) in ("Col", "SchemaCol"):
                idx = names.index(bare.name) if bare.name in names else -1
                if idx < 0:
                    raise ValueError(f'Column "{bare.name}" not found in schema {merged_schema}')
                if idx == 0:
                    outs.append((0, 0, key_kind))
                else:
                    outs.append((1, col_fold[idx - 1], stored(merged_kinds[idx], out_schema[o][1])))
                continue
            if b.string_tag(bare):
                raise FinishUnsupported("string expression after the merge")
            if n_prog >= hs.HS_MAX_OUTS:
                raise FinishUnsupported("too many computed columns for the fused tail")
            tag = b.emit_out(n_prog, col)
            if tag == "B":
                raise AssertionError("a comparison cannot be selected as a column (the reference has no BOOL type)")
            fin.prog_out[n_prog] = o
            outs.append((2, n_prog, stored(hs.F64 if tag == "F" else hs.I64, out_schema[o][1])))
            n_prog += 1
        if n_prog:
            lowered = b.finish()
            for slot, idx in enumerate(lowered.columns):
                if idx == 0 and key_kind == hs.STR:
                    raise FinishUnsupported("expression over a string key after the merge")
                fin.prog_src[slot] = -1 if idx == 0 else col_fold[idx - 1]
            prog = lowered.to_struct()
    if len(outs) > hs.HS_FINISH_MAX_OUT:
        raise FinishUnsupported("too many result columns for the fused tail")
    fin.n_out = len(outs)
    for o, (src, index, kind) in enumerate(outs):
        fin.outs[o].src, fin.outs[o].index, fin.outs[o].kind = src, index, kind
    return fin, prog, outs
