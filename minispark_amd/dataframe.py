"""Query builder with the reference's public surface (API to match: src/mini_spark/dataframe.py:28-86).

A ``DataFrame`` is a cursor over a growing chain of :mod:`minispark_amd.tasks` nodes; every builder call wraps the
current chain in one more node and hands the same object back, so calls chain.  What each call appends is declared
in ``_APPENDERS`` (method name -> node class + how the call's arguments map to the node's fields) and the methods
are generated from that table; only the calls that do more than append (``alias``, ``join``, running the query)
are written out.  Without an explicit engine the frame binds to the HIP engine on first use - this package has no
interpreted CPU engine.
"""

from __future__ import annotations

import copy
from pathlib import Path
from typing import Any, Callable

from . import tasks as _t
from .plan import PhysicalPlan

# builder call -> (task class, arguments -> keyword fields of that class)
_APPENDERS: dict[str, tuple[type, Callable[..., dict[str, Any]]]] = {
    "table": (_t.LoadTableBlockTask, lambda file_path: {"file_path": Path(file_path)}),
    "select": (_t.ProjectTask, lambda *columns: {"columns": list(columns)}),
    "filter": (_t.FilterTask, lambda column: {"condition": column}),
}


class GroupedData:
    """``df.group_by(col)``: waits for ``agg(...)`` to become an aggregate node over ``df``'s chain."""

    def __init__(self, df: "DataFrame", column: Any) -> None:
        self.df, self.group_column = df, column

    def agg(self, *agg_columns: Any) -> "DataFrame":
        return self.df._append(_t.AggregateTask, group_by_column=self.group_column, agg_columns=list(agg_columns))


class DataFrame:
    def __init__(self, engine: Any = None) -> None:
        self._engine = engine
        self.task: _t.Task = _t.VoidTask()

    # ---- chain construction ------------------------------------------------------------------------------
    def _append(self, node_class: type, **fields: Any) -> "DataFrame":
        self.task = node_class(self.task, **fields)
        return self

    def alias(self, alias_name: str) -> "DataFrame":
        if type(self.task) is not _t.LoadTableBlockTask:
            raise AssertionError("Alias can only be applied to table")
        self.task.alias = alias_name
        return self

    def group_by(self, column: Any) -> GroupedData:
        return GroupedData(self, column)

    def join(self, other_df: "DataFrame", on: Any, how: _t.JoinType) -> "DataFrame":
        return self._append(_t.BroadcastHashJoinTask, right_side_task=other_df.task, join_condition=on, how=how)

    # ---- binding + execution ---------------------------------------------------------------------------------
    @property
    def engine(self) -> Any:
        if self._engine is None:
            from .execution import HipExecutionEngine  # noqa: PLC0415 - loads libhipspark.so, needs a GPU

            self._engine = HipExecutionEngine()
        return self._engine

    @engine.setter
    def engine(self, engine: Any) -> None:
        self._engine = engine

    @property
    def schema(self) -> Any:
        return self.task.validate_schema()

    def _rows(self, limit: float | None = None) -> list:
        engine = self.engine
        results = engine.execute_full_task(self.task)
        rows = engine.collect_results(results) if limit is None else engine.collect_results(results, limit=limit)
        return list(rows)

    def collect(self) -> list:
        return self._rows()

    def collect_columns(self) -> dict:
        """The result column-wise: name -> numpy array (or list of str) - see ExecutionEngine.collect_columns."""
        engine = self.engine
        return engine.collect_columns(engine.execute_full_task(self.task))

    def show(self, n: int = 10) -> int:
        import tabulate  # noqa: PLC0415

        rows = self._rows(limit=n)
        print(tabulate.tabulate(rows, headers="keys", tablefmt="rounded_outline"))  # noqa: T201
        return len(rows)

    def explain(self, *, full: bool = False) -> None:
        snapshot = copy.deepcopy(self.task)  # planning annotates the nodes it is given
        print("Logical Plan")  # noqa: T201
        snapshot.explain()
        if full:
            PhysicalPlan.generate_physical_plan(snapshot).explain()


def _make_appender(name: str, node_class: type, to_fields: Callable[..., dict[str, Any]]) -> Callable[..., DataFrame]:
    def method(self: DataFrame, *args: Any, **kwargs: Any) -> DataFrame:
        return self._append(node_class, **to_fields(*args, **kwargs))

    method.__name__ = method.__qualname__ = name
    method.__doc__ = f"Wrap the chain in a {node_class.__name__}."
    return method


for _name, (_cls, _to_fields) in _APPENDERS.items():
    setattr(DataFrame, _name, _make_appender(_name, _cls, _to_fields))
