"""Fluent query builder over :mod:`minispark_amd.tasks` (reference: src/mini_spark/dataframe.py:28-86).

Only the subset the hot path's harness needs: ``table / alias / select / filter / group_by().agg /
join / collect / show / explain``, same method names and argument meaning as the reference.  The
default engine is the HIP engine - there is no interpreted CPU engine in this package.
"""

from __future__ import annotations

from copy import deepcopy
from pathlib import Path
from typing import TYPE_CHECKING

from .plan import PhysicalPlan
from .tasks import (
    AggregateTask,
    BroadcastHashJoinTask,
    FilterTask,
    JoinType,
    LoadTableBlockTask,
    ProjectTask,
    Task,
    VoidTask,
)

if TYPE_CHECKING:
    from .constants import Row, Schema
    from .execution import ExecutionEngine
    from .sql import AggCol, Col


class GroupedData:
    def __init__(self, df: "DataFrame", column: "Col") -> None:
        self.df = df
        self.group_column = column

    def agg(self, *agg_columns: "AggCol") -> "DataFrame":
        self.df.task = AggregateTask(self.df.task, group_by_column=self.group_column, agg_columns=list(agg_columns))
        return self.df


class DataFrame:
    def __init__(self, engine: "ExecutionEngine | None" = None) -> None:
        self._engine = engine
        self.task: Task = VoidTask()

    @property
    def engine(self) -> "ExecutionEngine":
        if self._engine is None:
            from .execution import HipExecutionEngine  # noqa: PLC0415 - loads the HIP library on first use

            self._engine = HipExecutionEngine()
        return self._engine

    @engine.setter
    def engine(self, engine: "ExecutionEngine") -> None:
        self._engine = engine

    @property
    def schema(self) -> "Schema":
        return self.task.validate_schema()

    def table(self, file_path: str) -> "DataFrame":
        self.task = LoadTableBlockTask(self.task, file_path=Path(file_path))
        return self

    def alias(self, alias_name: str) -> "DataFrame":
        if type(self.task) is not LoadTableBlockTask:
            raise AssertionError("Alias can only be applied to table")
        self.task.alias = alias_name
        return self

    def select(self, *columns: "Col") -> "DataFrame":
        self.task = ProjectTask(self.task, columns=list(columns))
        return self

    def filter(self, column: "Col") -> "DataFrame":
        self.task = FilterTask(self.task, condition=column)
        return self

    def group_by(self, column: "Col") -> GroupedData:
        return GroupedData(self, column)

    def join(self, other_df: "DataFrame", on: "Col", how: JoinType) -> "DataFrame":
        self.task = BroadcastHashJoinTask(self.task, right_side_task=other_df.task, join_condition=on, how=how)
        return self

    def collect(self) -> "list[Row]":
        job_results = self.engine.execute_full_task(self.task)
        return list(self.engine.collect_results(job_results))

    def show(self, n: int = 10) -> int:
        from tabulate import tabulate  # noqa: PLC0415

        results = self.engine.execute_full_task(self.task)
        rows = list(self.engine.collect_results(results, limit=n))
        print(tabulate(rows, tablefmt="rounded_outline", headers="keys"))  # noqa: T201
        return len(rows)

    def explain(self, *, full: bool = False) -> None:
        task = deepcopy(self.task)
        print("Logical Plan")  # noqa: T201
        task.explain()
        if full:
            PhysicalPlan.generate_physical_plan(task).explain()
