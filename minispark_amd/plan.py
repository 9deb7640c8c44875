"""Physical planning: logical task chain -> barrier-separated stages (reference: src/mini_spark/plan.py).

This is the build's own counterpart of the *caller* of the hot path: it decides
what the units of work are, and those units are part of the result semantics:

* one unit of partial aggregation per **file block** (``LoadTableBlockTask`` producer, plan.py:90-93);
* after a join, one unit per **shuffle partition** (``BroadcastHashJoinTask`` producer, plan.py:99-109);
* all partial rows of a key meet in exactly one final-merge unit (``LoadShuffleFilesTask``, plan.py:94-98).

Rewrites applied, in the reference's order (plan.py:224-235):

1. wrap the chain in ``WriteToLocalFileTask``; infer schemas bottom-up;
2. ``expand``: AggregateTask -> [Aggregate(before_shuffle) -> WriteToShufflePartitions(key) ->
   LoadShuffleFiles -> Aggregate(after)] (+ ProjectTask computing AVG = sum/count, plan.py:190-203);
   join inputs each get a WriteToShufflePartitions on their key (plan.py:186-189);
3. strip ``alias.`` prefixes from the output column names with a final ProjectTask (plan.py:207-222);
4. cut the chain into stages at every shuffle write / join, dependencies first.
"""

from __future__ import annotations

from copy import deepcopy

from .sql import AggCol, Col
from .tasks import (
    AggregateTask,
    BroadcastHashJoinTask,
    ConsumerTask,
    LoadShuffleFilesTask,
    ProducerTask,
    ProjectTask,
    Task,
    VoidTask,
    WriterTask,
    WriteToLocalFileTask,
    WriteToShufflePartitions,
)


class Stage:
    """producer -> consumers -> writer; ``dependencies`` are the stages whose shuffle output it reads
    (for a join: [left/build stage, right/probe stage])."""

    def __init__(self, full_task: Task, dependencies: list["Stage"]) -> None:
        self.full_task = full_task
        self.dependencies = dependencies
        self.stage_id = ""
        self.job_results: list = []
        chain = list(full_task.task_chain)
        producer, consumers, writer = chain[0], chain[1:-1], chain[-1]
        if not isinstance(producer, ProducerTask):
            raise AssertionError(f"stage must start with a producer, got {type(producer).__name__}")
        if not all(isinstance(c, ConsumerTask) for c in consumers):
            raise AssertionError("stage interior must be consumers")
        if not isinstance(writer, WriterTask):
            raise AssertionError(f"stage must end with a writer, got {type(writer).__name__}")
        self.producer: ProducerTask = producer
        self.consumers: list[ConsumerTask] = consumers  # type: ignore[assignment]
        self.writer: WriterTask = writer

    def late_initialize(self, stage_id: str) -> None:
        self.stage_id = stage_id

    def explain(self) -> None:
        self.full_task.explain()

    def __repr__(self) -> str:
        consumers = f"[{','.join(type(c).__name__ for c in self.consumers)}] -> " if self.consumers else ""
        deps = ",".join(str(d.stage_id) for d in self.dependencies)
        return (
            f"[Stage {self.stage_id}: {type(self.producer).__name__} -> {consumers}"
            f"{type(self.writer).__name__}, deps: ({deps})]"
        )


def _cut(top: Task) -> Stage:
    """Detach ``top``'s chain at the first shuffle boundary below it and build stages recursively.

    ``top`` is always a writer.  Walking down from it, the stage ends either at a join (whose two
    inputs are separate stages) or where the parent is a WriteToShufflePartitions (which tops the
    next stage down)."""
    node = top
    while True:
        if type(node) is BroadcastHashJoinTask:
            left_top, right_top = node.parent_task, node.right_side_task
            node.parent_task, node.right_side_task = VoidTask(), VoidTask()
            return Stage(top, [_cut(left_top), _cut(right_top)])
        parent = node.parent_task
        if parent is None or type(parent) is VoidTask:
            return Stage(top, [])
        if type(parent) is WriteToShufflePartitions:
            node.parent_task = VoidTask()
            return Stage(top, [_cut(parent)])
        node = parent


def _execution_order(stage: Stage, out: list[Stage]) -> None:
    for dep in stage.dependencies:
        _execution_order(dep, out)
    out.append(stage)


class PhysicalPlan:
    def __init__(self, stages: list[Stage]) -> None:
        self.stages = stages

    @staticmethod
    def infer_schema(task: Task) -> None:
        node: Task | None = task
        while node is not None and type(node) is not VoidTask:
            node.inferred_schema = node.validate_schema()
            if type(node) is BroadcastHashJoinTask:
                PhysicalPlan.infer_schema(node.right_side_task)
            node = node.parent_task

    @staticmethod
    def expand_tasks(task: Task) -> Task:
        if type(task) is VoidTask:
            return task
        task.parent_task = PhysicalPlan.expand_tasks(task.parent_task)
        if type(task) is BroadcastHashJoinTask:
            task.right_side_task = PhysicalPlan.expand_tasks(task.right_side_task)
            task.parent_task = WriteToShufflePartitions(task.parent_task, key_column=task.left_key)
            task.right_side_task = WriteToShufflePartitions(task.right_side_task, key_column=task.right_key)
            return task
        if type(task) is AggregateTask and task.before_shuffle:
            requested = task.agg_columns
            carried = [part for agg in requested for part in agg.expand_avg()]
            partial = AggregateTask(task.parent_task, group_by_column=task.group_by_column, agg_columns=carried)
            shuffled = WriteToShufflePartitions(partial, key_column=task.group_by_column)
            task.parent_task = LoadShuffleFilesTask(shuffled)
            task.before_shuffle = False
            task.agg_columns = [AggCol(agg.type, Col(agg.name)) for agg in carried]
            if any(agg.type == "avg" for agg in requested):
                return ProjectTask(
                    task, columns=[Col(task.group_by_column.name), *[agg.projection() for agg in requested]]
                )
        return task

    @staticmethod
    def cleanup_output_column_names(task: Task) -> None:
        schema = task.inferred_schema
        if schema is None:
            raise AssertionError("schema not inferred")
        if not any("." in name for name, _ in schema):
            return
        renamed = [Col(name).alias(name.rsplit(".", 1)[-1]) for name, _ in schema]
        task.parent_task = ProjectTask(task.parent_task, columns=renamed)
        clean = [(col.name, col_type) for col, (_, col_type) in zip(renamed, schema, strict=True)]
        task.parent_task.inferred_schema = clean
        task.inferred_schema = clean

    @staticmethod
    def generate_physical_plan(full_task: Task) -> "PhysicalPlan":
        # planning rewrites nodes in place; work on a copy so the caller's DataFrame stays reusable
        root: Task = WriteToLocalFileTask(deepcopy(full_task))
        PhysicalPlan.infer_schema(root)
        root = PhysicalPlan.expand_tasks(root)
        PhysicalPlan.infer_schema(root)
        PhysicalPlan.cleanup_output_column_names(root)
        stages: list[Stage] = []
        _execution_order(_cut(root), stages)
        for i, stage in enumerate(stages):
            stage.late_initialize(str(i))
        return PhysicalPlan(stages)

    def explain(self) -> None:
        for i, stage in enumerate(self.stages):
            print("Stage", i)  # noqa: T201
            stage.explain()
            print("-" * 10)  # noqa: T201
