"""ORACLE (test infrastructure - never imported by the product).

Row-at-a-time CPU restatement of the reference's ``PythonExecutionEngine`` for the hot path
(/root/reference/src/mini_spark/): every function cites the reference lines it follows.  Pinned
against the real reference by tests/golden/*.json (made by tests/golden/make_golden.py, which imports
the reference in the build container) - see tests/test_oracle_golden.py.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may use this module,
and only as the checker.

What is restated, with the quantisation points that make it the *reference's* result and not merely a
correct SQL answer:

* per job (one file block, or one join partition) accumulate in row order in Python float / int;
* at the stage boundary every value goes through the shuffle file: FLOAT -> f32, INTEGER -> i32
  (OverflowError outside the range), TIMESTAMP -> microseconds;
* the final stage merges partial rows per key in shuffle-file order in fp64, AVG = sum / count on the
  un-rounded merged sum, and the result file rounds to f32 again.

Deliberate divergence (documented in DESIGN.md): the join uses partition-global left row indices - the
reference indexes per chunk (tasks.py:217), which is only correct while a left partition fits one block.
"""

from __future__ import annotations

import operator
import re
from datetime import datetime
from pathlib import Path
from typing import Any, Callable, Iterable

from . import blockfile as bfio

MAX_INT = 2**31 - 1  # constants.py:14
MIN_INT = -(2**31)  # constants.py:15
SHUFFLE_PARTITIONS = 10  # constants.py:8

_TYPE_CODE = {"INTEGER": bfio.INTEGER, "STRING": bfio.STRING, "FLOAT": bfio.FLOAT, "TIMESTAMP": bfio.TIMESTAMP}


def _cls(x: Any) -> str:
    return type(x).__name__


# --------------------------------------------------------------------------------------------------
# expression evaluation: sql.py:127-128 (SchemaCol), :147-148 (Alias), :192-194 (LIKE), :262-266 (binary),
# :371-372 (Lit), :417-418 (AggCol delegates to its argument)
# --------------------------------------------------------------------------------------------------
_OPS: dict[str, Callable[[Any, Any], Any]] = {
    name: getattr(operator, name)
    for name in ("add", "sub", "mul", "truediv", "floordiv", "mod", "eq", "ne", "lt", "le", "gt", "ge", "and_", "or_")
}


def compile_expr(expr: Any, schema: list[tuple[str, Any]]) -> Callable[[tuple], Any]:
    """schema_executor + execute_row: returns row -> value (rows are tuples in schema order)."""
    kind = _cls(expr)
    if kind in ("AliasColumn", "AggCol"):
        return compile_expr(expr.original_col, schema)
    if kind in ("Col", "SchemaCol"):
        names = [n for n, _ in schema]
        if expr.name not in names:
            raise ValueError(f"Column {expr.name} not found in schema {schema}")
        pos = names.index(expr.name)
        return lambda row: row[pos]
    if kind == "Lit":
        value = expr.value
        return lambda row: value
    if kind == "LikeColumn":
        inner = compile_expr(expr.original_col, schema)
        # sql.py:178-179
        regex = re.compile("^" + re.escape(expr.pattern).replace("%", ".*").replace("_", ".") + "$")
        return lambda row: regex.match(str(inner(row))) is not None
    if kind == "BinaryOperatorColumn":
        left = compile_expr(expr.left_side, schema)
        right = compile_expr(expr.right_side, schema)
        fn = _OPS[expr.operator.__name__]
        return lambda row: fn(left(row), right(row))
    raise NotImplementedError(kind)


def project_column(expr: Any, chunk: list[list[Any]], schema: list[tuple[str, Any]]) -> list[Any]:
    fn = compile_expr(expr, schema)  # tasks.py:32-35
    return [fn(row) for row in zip(*chunk)] if chunk and chunk[0] else []


# --------------------------------------------------------------------------------------------------
# shuffle "files": rows that went through BlockFile.append_tuples + a later read (io.py:254-256, 87-109, 129-149)
# --------------------------------------------------------------------------------------------------
def quantise_column(col_type_name: str, values: list[Any]) -> list[Any]:
    """What a column looks like after being written to and read back from a BlockFile."""
    if col_type_name == "FLOAT":
        for v in values:
            assert type(v) is float, f"FLOAT column holds {type(v).__name__}"  # io.py:93
        return [bfio.f32(v) for v in values]
    if col_type_name == "INTEGER":
        out = []
        for v in values:
            assert type(v) is int, f"INTEGER column holds {type(v).__name__}"  # io.py:89
            v.to_bytes(4, byteorder="little", signed=True)  # raises OverflowError like io.py:90
            out.append(v)
        return out
    if col_type_name == "TIMESTAMP":
        return [bfio.from_us(bfio.to_us(datetime.fromisoformat(v) if type(v) is str else v)) for v in values]
    if col_type_name == "STRING":
        for v in values:
            assert type(v) is str
        return list(values)
    raise ValueError(col_type_name)


class ShuffleStore:
    """partition -> list of row tuples, in append order (all jobs of a stage append to the same
    ``shuffle/<stage>/<partition>.bin``, tasks.py:366-375)."""

    def __init__(self) -> None:
        self.partitions: dict[int, list[tuple]] = {}

    def append(self, partition: int, rows: list[tuple]) -> None:
        self.partitions.setdefault(partition, []).extend(rows)


# --------------------------------------------------------------------------------------------------
# operators
# --------------------------------------------------------------------------------------------------
def filter_chunk(task: Any, chunk: list[list[Any]], schema: list) -> list[list[Any]]:
    cond = project_column(task.condition, chunk, schema)  # tasks.py:167-177
    return [[v for v, c in zip(col, cond) if c] for col in chunk]


def project_chunk(task: Any, chunk: list[list[Any]], schema: list) -> list[list[Any]]:
    return [project_column(col, chunk, schema) for col in task.columns]  # tasks.py:80-85


def fill_aggregators(aggregators: list[dict], agg_columns: list[Any], group_column: list[Any],
                     agg_expr_columns: list[list[Any]]) -> None:
    """tasks.py:295-310."""
    for counter, agg_col, values in zip(aggregators, agg_columns, agg_expr_columns):
        if agg_col.type == "sum":
            for group, x in zip(group_column, values):
                assert type(x) in (int, float)
                counter[group] = counter.get(group, 0) + x
        if agg_col.type == "min":
            for group, x in zip(group_column, values):
                assert type(x) in (int, float)
                counter[group] = min(counter.get(group, MAX_INT), x)
        if agg_col.type == "max":
            for group, x in zip(group_column, values):
                assert type(x) in (int, float)
                counter[group] = max(counter.get(group, MIN_INT), x)


def emit_aggregators(aggregators: list[dict]) -> list[list[Any]]:
    all_keys = list({key for agg in aggregators for key in agg})  # tasks.py:272-278
    return [all_keys, *[[agg.get(key, 0) for key in all_keys] for agg in aggregators]]


def python_partition(key: Any) -> int:
    return hash(key) % SHUFFLE_PARTITIONS  # tasks.py:362


def type_name(col_type: Any) -> str:
    return getattr(col_type, "name", str(col_type))


# --------------------------------------------------------------------------------------------------
# engine
# --------------------------------------------------------------------------------------------------
class OracleEngine:
    """Sequential interpretation of a physical plan: for each stage, for each job, run the pipeline
    (execution.py:69-83, plan.py:70-87)."""

    def __init__(self, rows_per_block: int | None = None) -> None:
        self.rows_per_block = rows_per_block

    def run(self, plan: Any) -> list[dict]:
        """Execute ``plan.stages``; returns the result rows as they would be read back from the
        result file (FLOAT columns rounded to f32)."""
        stores: dict[int, ShuffleStore] = {}
        result_rows: list[dict] = []
        for stage in plan.stages:
            store = ShuffleStore()
            stores[id(stage)] = store
            for chunks in self._jobs(stage, stores):
                rows = self._run_job(stage, chunks, store)
                result_rows.extend(rows)
        return result_rows

    # ---- job creation: plan.py:89-111 ----------------------------------------------------------------
    def _jobs(self, stage: Any, stores: dict[int, ShuffleStore]) -> Iterable[Iterable[list[list[Any]]]]:
        producer = stage.producer
        kind = _cls(producer)
        if kind == "LoadTableBlockTask":
            _, blocks = bfio.read_blockfile(Path(producer.file_path))  # one ScanJob per block
            for block in blocks:
                yield [block]
        elif kind == "LoadShuffleFilesTask":
            dep = stores[id(stage.dependencies[0])]
            for rows in dep.partitions.values():  # one job per partition
                yield self._blocks(rows, len(producer.inferred_schema))
        elif kind == "BroadcastHashJoinTask":
            left = stores[id(stage.dependencies[0])]
            right = stores[id(stage.dependencies[1])]
            for partition in set(left.partitions) | set(right.partitions):
                yield self._join(producer, left.partitions.get(partition, []), right.partitions.get(partition, []))
        else:
            raise NotImplementedError(kind)

    def _blocks(self, rows: list[tuple], ncols: int) -> list[list[list[Any]]]:
        """A shuffle file is read back block by block (tasks.py:144-150); the block size only matters
        for chunking, never for values."""
        from minispark_amd import constants  # noqa: PLC0415 - the knob tests patch

        per = self.rows_per_block or constants.ROWS_PER_BLOCK
        out = []
        for lo in range(0, len(rows), per):
            part = rows[lo : lo + per]
            out.append([list(col) for col in zip(*part)] if part else [[] for _ in range(ncols)])
        return out

    def _join(self, task: Any, left_rows: list[tuple], right_rows: list[tuple]) -> Iterable[list[list[Any]]]:
        """tasks.py:201-240: hash map over the whole left partition, stream the right partition."""
        left_schema, right_schema = task.left_schema, task.right_schema
        left_cols = [list(c) for c in zip(*left_rows)] if left_rows else [[] for _ in left_schema]
        key_fn = compile_expr(task.left_key, left_schema)
        row_map: dict[Any, list[int]] = {}
        for idx, row in enumerate(left_rows):  # partition-global index (see module docstring)
            row_map.setdefault(key_fn(row), []).append(idx)
        rkey_fn = compile_expr(task.right_key, right_schema)
        ncols = len(left_schema) + len(right_schema)
        for chunk in self._blocks(right_rows, len(right_schema)):
            out: list[list[Any]] = [[] for _ in range(ncols)]
            for rrow in zip(*chunk):
                for lidx in row_map.get(rkey_fn(rrow), []):
                    for i, col in enumerate(left_cols):
                        out[i].append(col[lidx])
                    for j, v in enumerate(rrow):
                        out[len(left_cols) + j].append(v)
            yield out

    # ---- pipeline: plan.py:70-87 -----------------------------------------------------------------------
    def _run_job(self, stage: Any, chunks: Iterable[list[list[Any]]], store: ShuffleStore) -> list[dict]:
        aggregators: dict[int, list[dict]] = {}
        results: list[dict] = []

        def push(chunk: list[list[Any]] | None, is_last: bool) -> None:
            schema = stage.producer.inferred_schema
            for task in stage.consumers:
                kind = _cls(task)
                if kind == "AggregateTask":
                    if is_last and chunk is None:
                        aggs = aggregators.get(id(task))
                        chunk = emit_aggregators(aggs) if aggs is not None else [[] for _ in task.inferred_schema]
                    else:
                        aggs = aggregators.setdefault(id(task), [{} for _ in task.agg_columns])
                        if task.before_shuffle:  # tasks.py:284-289
                            group = project_column(task.group_by_column, chunk, schema)
                            values = [project_column(a, chunk, schema) for a in task.agg_columns]
                            fill_aggregators(aggs, task.agg_columns, group, values)
                        else:  # tasks.py:290-292
                            fill_aggregators(aggs, task.agg_columns, chunk[0], chunk[1:])
                        chunk = None
                elif chunk is not None:
                    if kind == "FilterTask":
                        chunk = filter_chunk(task, chunk, schema)
                    elif kind == "ProjectTask":
                        chunk = project_chunk(task, chunk, schema)
                    else:
                        raise NotImplementedError(kind)
                schema = task.inferred_schema
            if chunk is not None:
                results.extend(self._write(stage, chunk, store))

        for chunk in chunks:
            push(chunk, False)
        push(None, True)
        return results

    # ---- writers ---------------------------------------------------------------------------------------
    def _write(self, stage: Any, chunk: list[list[Any]], store: ShuffleStore) -> list[dict]:
        writer = stage.writer
        schema = writer.inferred_schema
        if len(chunk) == 0 or len(chunk[0]) == 0:
            return []
        quantised = [quantise_column(type_name(t), col) for (_, t), col in zip(schema, chunk)]
        if _cls(writer) == "WriteToShufflePartitions":  # tasks.py:347-375
            key_schema = writer.parent_task.inferred_schema
            keys = project_column(writer.key_column, chunk, key_schema)
            buckets: dict[int, list[tuple]] = {}
            for row, key in zip(zip(*quantised), keys):
                buckets.setdefault(python_partition(key), []).append(row)
            for partition in sorted(buckets):
                store.append(partition, buckets[partition])
            return []
        if _cls(writer) == "WriteToLocalFileTask":  # tasks.py:400-410
            names = [n for n, _ in schema]
            return [dict(zip(names, row)) for row in zip(*quantised)]
        raise NotImplementedError(_cls(writer))


def run_query(task: Any, rows_per_block: int | None = None) -> list[dict]:
    """Plan (with the product's planner, itself pinned against the reference's plans by the golden
    fixtures) and execute on the oracle."""
    from minispark_amd.plan import PhysicalPlan  # noqa: PLC0415

    plan = PhysicalPlan.generate_physical_plan(task)
    return OracleEngine(rows_per_block).run(plan)

