"""Logical / physical operator nodes of a query (reference: src/mini_spark/tasks.py).

Host-side mirror of the reference's ``Task`` classes: same class names, same fields, same schema
rules and error messages - so a plan built by the reference lowers through the same code as one built
here (:mod:`minispark_amd.lowering` dispatches on class names).  Unlike the reference the nodes carry
**no row-processing code**: the reference's ``generate_chunks`` / ``execute`` / ``write`` bodies
(tasks.py:117-121,167-177,201-240,270-310,347-375,400-410) are what the HIP kernels replace.

Nodes form a linked list through ``parent_task`` (a join also has ``right_side_task``); ``VoidTask``
terminates the list.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path
from typing import Iterator, Literal

from .constants import Schema
from .io import BlockFile
from .sql import AggCol, BinaryOperatorColumn, Col, LikeColumn

JoinType = Literal["inner", "left", "right", "outer"]


def nice_schema(schema: Schema | None) -> str:
    if schema is None:
        return ""
    return "[" + ", ".join(f"{name}:{col_type}" for name, col_type in schema) + "]"


def _plain_column_names(*exprs: Col) -> list[str]:
    return [c.name for e in exprs for c in e.all_nested_columns if type(c).__name__ == "Col"]


def _check_known(exprs: list[Col], schema: Schema, where: str) -> None:
    known = {name for name, _ in schema}
    unknown = [n for n in dict.fromkeys(_plain_column_names(*exprs)) if n not in known]
    if unknown:
        raise ValueError(f"Unknown columns in {where}: {unknown}")


@dataclass
class Task:
    parent_task: "Task" = field(repr=False)
    inferred_schema: Schema | None = None

    def validate_schema(self) -> Schema:
        return self.parent_task.validate_schema()

    @property
    def task_chain(self) -> Iterator["Task"]:
        if type(self) is VoidTask:
            return
        yield from self.parent_task.task_chain
        yield self

    def describe(self) -> str:
        return type(self).__name__

    def explain(self, lvl: int = 0) -> None:
        indent = "  " * lvl + ("+- " if lvl > 0 else "")
        print(f"{indent} {self.describe()}:{nice_schema(self.inferred_schema)}")  # noqa: T201
        self.parent_task.explain(lvl + 1)


@dataclass
class VoidTask(Task):
    parent_task: Task | None = None  # type: ignore[assignment]

    def validate_schema(self) -> Schema:
        return []

    def explain(self, lvl: int = 0) -> None:
        pass


@dataclass(kw_only=True)
class ProducerTask(Task):
    pass


@dataclass(kw_only=True)
class ConsumerTask(Task):
    pass


@dataclass(kw_only=True)
class WriterTask(Task):
    pass


@dataclass(kw_only=True)
class LoadTableBlockTask(ProducerTask):
    """Scan of one BlockFile; one job per file block (tasks.py:112-138)."""

    file_path: Path
    alias: str = ""

    @property
    def file_schema(self) -> Schema:
        return BlockFile(self.file_path).file_schema

    def validate_schema(self) -> Schema:
        if self.parent_task.validate_schema() != []:
            raise AssertionError("a table scan has no input")
        if not self.alias:
            return list(self.file_schema)
        return [(f"{self.alias}.{name}", col_type) for name, col_type in self.file_schema]

    def describe(self) -> str:
        return f"LoadTableBlockTask({self.file_path})"


@dataclass(kw_only=True)
class LoadShuffleFilesTask(ProducerTask):
    """Reads back what the previous stage shuffled; one job per partition (tasks.py:141-156)."""

    def describe(self) -> str:
        return "LoadShuffleFile()"


@dataclass(kw_only=True)
class ProjectTask(ConsumerTask):
    columns: list[Col]

    def validate_schema(self) -> Schema:
        schema = self.parent_task.validate_schema()
        expanded: list[Col] = []
        for col in self.columns:
            if type(col).__name__ == "Col" and col.name == "*":
                expanded.extend(Col(name) for name, _ in schema)
            else:
                expanded.append(col)
        self.columns = expanded
        _check_known(self.columns, schema, "projection")
        return [(col.name, col.infer_type(schema)) for col in self.columns]

    def describe(self) -> str:
        return f"Project({', '.join(str(c) for c in self.columns)})"


@dataclass(kw_only=True)
class FilterTask(ConsumerTask):
    condition: Col

    def __post_init__(self) -> None:
        if type(self.condition).__name__ not in {"BinaryOperatorColumn", "LikeColumn"}:
            raise AssertionError(type(self.condition))

    def validate_schema(self) -> Schema:
        schema = self.parent_task.validate_schema()
        self.condition.infer_type(schema)
        return schema

    def describe(self) -> str:
        return f"Filter({self.condition})"


@dataclass(kw_only=True)
class BroadcastHashJoinTask(ProducerTask):
    """Inner equi-join on one column pair.  Despite the name both inputs are hash-partitioned on the
    key; the LEFT input (``parent_task``) is the build side (tasks.py:190-260)."""

    right_side_task: Task
    join_condition: Col
    how: JoinType = "inner"
    left_key: Col | None = None
    right_key: Col | None = None
    left_schema: Schema | None = None
    right_schema: Schema | None = None

    def validate_schema(self) -> Schema:
        self.left_schema = self.parent_task.validate_schema()
        self.right_schema = self.right_side_task.validate_schema()
        _check_known([self.join_condition], self.left_schema + self.right_schema, "Join")
        if type(self.join_condition).__name__ != "BinaryOperatorColumn":
            raise AssertionError("Only equi-join is supported")
        self.left_key, self.right_key = self.join_condition.extract_left_right_key(
            self.left_schema, self.right_schema
        )
        return self.left_schema + self.right_schema

    def describe(self) -> str:
        return f'Join({self.join_condition}, "{self.how}")'

    def explain(self, lvl: int = 0) -> None:
        indent = "  " * lvl + ("+- " if lvl > 0 else "")
        print(f"{indent} {self.describe()}:{nice_schema(self.inferred_schema)}")  # noqa: T201
        self.parent_task.explain(lvl + 1)
        self.right_side_task.explain(lvl + 1)


@dataclass(kw_only=True)
class AggregateTask(ConsumerTask):
    """Hash group-by on ONE plain column.  ``before_shuffle`` = the per-job partial phase, otherwise
    the merge phase that combines column i+1 of the shuffled partial rows with aggregate i
    (tasks.py:263-340)."""

    group_by_column: Col
    agg_columns: list[AggCol]
    before_shuffle: bool = True

    def validate_schema(self) -> Schema:
        schema = self.parent_task.validate_schema()
        if not self.before_shuffle:
            return schema
        _check_known([*self.agg_columns, self.group_by_column], schema, "aggregation")
        return [
            (self.group_by_column.name, self.group_by_column.infer_type(schema)),
            *[(agg.name, agg.infer_type(schema)) for agg in self.agg_columns],
        ]

    def describe(self) -> str:
        return (
            f"AggregateTask(group_by: {self.group_by_column}, agg: {self.agg_columns}, "
            f"before_shuffle:{self.before_shuffle})"
        )


@dataclass
class WriteToShufflePartitions(WriterTask):
    """Stage boundary: rows are routed by ``hash(key) % SHUFFLE_PARTITIONS`` and quantised to the
    on-disk types (FLOAT->f32, INTEGER->i32) - tasks.py:343-397."""

    key_column: Col | None = None

    def validate_schema(self) -> Schema:
        schema = self.parent_task.validate_schema()
        if self.key_column is None:
            return schema
        known = {name for name, _ in schema}
        unknown = [n for n in _plain_column_names(self.key_column) if n not in known]
        if unknown:
            raise ValueError(f"Unknown columns in GroupBy: {unknown}")
        if self.key_column.name in known:
            return schema
        return [(self.key_column.name, self.key_column.infer_type(schema)), *schema]

    def describe(self) -> str:
        return f"WriteToShufflePartitions({self.key_column})"


@dataclass(kw_only=True)
class WriteToLocalFileTask(WriterTask):
    def describe(self) -> str:
        return "WriteToLocalFileTask()"


__all__ = [
    "AggregateTask",
    "BinaryOperatorColumn",
    "BroadcastHashJoinTask",
    "ConsumerTask",
    "FilterTask",
    "JoinType",
    "LikeColumn",
    "LoadShuffleFilesTask",
    "LoadTableBlockTask",
    "ProducerTask",
    "ProjectTask",
    "Task",
    "VoidTask",
    "WriteToLocalFileTask",
    "WriteToShufflePartitions",
    "WriterTask",
    "nice_schema",
]
