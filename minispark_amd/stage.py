"""Host side of the stage-level C ABI (include/hipspark.h, csrc/hs_engine.hip): lower a GROUP BY query into the plan
blob, and a thin handle over ``hs_engine_* / hs_table_* / hs_stage_* / hs_result_*``.

No torch, no :class:`minispark_amd.device.Device`: this is the whole host a cgo / JNI / FFI binding has to reproduce
 - everything else (BlockFile reading, buffers, geometry, retries, replay, hand-over) happens
behind the ABI.  The query shape it covers is the hot path's: ``table -> [filter]* -> group_by(col).agg(...)``, i.e. the
reference's two stages [Load -> Filter* -> Aggregate(before) -> shuffle] + [shuffle -> Aggregate(after) -> (Project) ->
result] (plan.py:182-204); round 3: a SELECT in front of the GROUP BY (its columns inlined, a computed INTEGER key
materialised by the library), and any number of groups per block the on-chip tiers hold.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Any

import numpy as np

from . import hipspark as hs
from .constants import ColumnType, Row, Schema
from .io import BlockFile, StrCol, rows_from_raw
from .lowering import ProgramBuilder, lower_aggregate, lower_finish, unalias
from .sql import Col

_FILE_KIND = {ColumnType.INTEGER: hs.I32, ColumnType.FLOAT: hs.F32, ColumnType.TIMESTAMP: hs.I64, ColumnType.STRING: hs.STR}
_TYPE_CODE = {ColumnType.INTEGER: 0, ColumnType.STRING: 1, ColumnType.FLOAT: 2, ColumnType.TIMESTAMP: 3}
_NP = {hs.I32: np.int32, hs.F32: np.float32, hs.I64: np.int64}


class StageUnsupported(NotImplementedError):
    """The query is not of the shape the stage-level path runs (the engine's general path takes it)."""


def _cls(obj: Any) -> str:
    return type(obj).__name__


def _substitute(node: Any, defs: dict[str, Any] | None) -> Any:
    """``node`` over the columns a ProjectTask produced -> the same expression over the table's columns (every projected
    name replaced by its definition): the projection is a pure function of the row (reference tasks.py:32-35,
    sql.py:262-266) and its values are not stored in between, so inlining it gives the same values."""
    if defs is None:
        return node
    name = _cls(node)
    if name == "AliasColumn":
        return _substitute(node.original_col, defs)
    if name in ("Col", "SchemaCol"):
        if node.name not in defs:
            raise ValueError(f'Column "{node.name}" not found in schema {list(defs)}')
        return defs[node.name]
    if name == "Lit":
        return node
    if name == "LikeColumn":
        return type(node)(_substitute(node.original_col, defs), node.pattern)
    if name == "BinaryOperatorColumn":
        return type(node)(_substitute(node.left_side, defs), _substitute(node.right_side, defs), node.operator)
    raise StageUnsupported(f"{name} over a projected column")


def _int_bits(node: Any, schema: Schema) -> int | None:
    """b with |value| <= 2**b for an INTEGER-valued expression that cannot raise on ANY row (no division by a column,
    no overflow of the 64-bit cells), else None.  A computed GROUP BY key is evaluated over all rows of the table -
    also the ones the WHERE drops - which is only the reference's behaviour when nothing can go wrong on those rows."""
    name = _cls(node)
    if name == "AliasColumn":
        return _int_bits(node.original_col, schema)
    if name in ("Col", "SchemaCol"):
        types = dict(schema)
        return 31 if types.get(node.name) == ColumnType.INTEGER else None
    if name == "Lit":
        return node.value.bit_length() if type(node.value) is int else None
    if name != "BinaryOperatorColumn":
        return None
    op = node.operator.__name__
    left = _int_bits(node.left_side, schema)
    if left is None:
        return None
    if op in ("mod", "floordiv"):
        right = unalias(node.right_side)
        if _cls(right) != "Lit" or type(right.value) is not int or right.value == 0:
            return None
        return abs(right.value).bit_length() if op == "mod" else left + 1
    right_bits = _int_bits(node.right_side, schema)
    if right_bits is None:
        return None
    bits = {"add": max(left, right_bits) + 1, "sub": max(left, right_bits) + 1, "mul": left + right_bits}.get(op)
    return bits if bits is not None and bits <= 62 else None


COMPUTED_KEY = "__hs_computed_key"


def lower_stage_plan(full_task: Any, plan: Any = None) -> tuple[hs.hs_stage_plan, Path, Schema]:
    """Task tree (reference or this package's classes) -> (plan blob, table path, result schema)."""
    if plan is None:
        from .plan import PhysicalPlan  # noqa: PLC0415

        plan = PhysicalPlan.generate_physical_plan(full_task)
    stages = list(plan.stages)
    if len(stages) != 2:
        raise StageUnsupported(f"{len(stages)} stages: the stage-level path runs scan + GROUP BY queries")
    scan, final = (stages if _cls(stages[0].producer) == "LoadTableBlockTask" else stages[::-1])
    if _cls(scan.producer) != "LoadTableBlockTask" or _cls(final.producer) != "LoadShuffleFilesTask":
        raise StageUnsupported("not a scan stage feeding a final stage")
    filters, partial = [], None
    defs: dict[str, Any] | None = None  # after a ProjectTask: projected name -> its expression over the table's columns
    for task in scan.consumers:
        if _cls(task) == "FilterTask" and partial is None:
            filters.append(_substitute(task.condition, defs))
        elif _cls(task) == "ProjectTask" and partial is None:
            names = [n for n, _ in task.inferred_schema]
            if len(names) != len(task.columns) or any(getattr(unalias(c), "name", "") == "*" for c in task.columns):
                raise StageUnsupported("SELECT * in the scan stage")
            defs = {n: _substitute(unalias(c), defs) for n, c in zip(names, task.columns)}
        elif _cls(task) == "AggregateTask" and task.before_shuffle and partial is None:
            partial = task
        else:
            raise StageUnsupported(f"{_cls(task)} in the scan stage")
    consumers = list(final.consumers)
    if partial is None or not consumers or _cls(consumers[0]) != "AggregateTask" or consumers[0].before_shuffle:
        raise StageUnsupported("no partial / final aggregate pair")
    if len(consumers) > 2 or (len(consumers) == 2 and _cls(consumers[1]) != "ProjectTask"):
        raise StageUnsupported("more than a projection after the final aggregate")
    merge = consumers[0]
    project = list(consumers[1].columns) if len(consumers) == 2 else None
    out_schema = list(final.writer.inferred_schema)

    table_schema = list(scan.producer.inferred_schema)
    prefix = f"{scan.producer.alias}." if getattr(scan.producer, "alias", "") else ""
    kinds = [_FILE_KIND[t] for _, t in table_schema]
    group_by, agg_columns = partial.group_by_column, list(partial.agg_columns)
    key_program = None
    if defs is not None:  # the projection, inlined (values are not stored between a ProjectTask and its consumer)
        import copy  # noqa: PLC0415

        group_by = _substitute(group_by, defs)
        for i, agg in enumerate(agg_columns):
            agg_columns[i] = copy.copy(agg)
            agg_columns[i].original_col = _substitute(agg.original_col, defs)
        if _cls(unalias(group_by)) not in ("Col", "SchemaCol"):
            # GROUP BY a computed column: materialised per run by the library as a stored INTEGER column next to the table's
            bits = _int_bits(group_by, table_schema)
            if bits is None or bits > 30:
                raise StageUnsupported("a computed GROUP BY key must be an INTEGER expression that provably fits the stored type")
            kb = ProgramBuilder(table_schema, kinds)
            if kb.emit_out(0, group_by) != "I":
                raise StageUnsupported("a computed GROUP BY key must be INTEGER-valued")
            key_program = kb.finish()
            table_schema = table_schema + [(COMPUTED_KEY, ColumnType.INTEGER)]
            kinds = kinds + [hs.I32]
            group_by = Col(COMPUTED_KEY)
    low = lower_aggregate(table_schema, kinds, filters, group_by, agg_columns)
    if low.numeric_slots > hs.HS_FUSED_COLS:
        raise StageUnsupported(f"more than {hs.HS_FUSED_COLS} numeric columns")
    acc_kinds = [hs.I32 if is_int else hs.F32 for is_int in low.acc_is_int]
    key_idx = low.program.columns[low.key_slot]
    fin, fin_prog, outs = lower_finish(low.agg_to_acc, acc_kinds, kinds[key_idx], merge.agg_columns, merge.inferred_schema,
                                       project, out_schema)
    blob = hs.hs_stage_plan()
    blob.version = hs.HS_STAGE_PLAN_VERSION
    blob.n_cols = len(low.program.columns)
    for slot, idx in enumerate(low.program.columns):
        blob.col_ids[slot] = -1 if key_program is not None and idx == len(table_schema) - 1 else idx
    if key_program is not None:
        if any(idx == len(table_schema) - 1 for s_, idx in enumerate(low.program.columns) if s_ != low.key_slot):
            raise AssertionError("the computed key column is read by name only as the key")
        blob.key_computed, blob.n_kcols = 1, len(key_program.columns)
        for slot, idx in enumerate(key_program.columns):
            blob.kcol_ids[slot] = idx
        blob.key_prog = key_program.to_struct()
    blob.key_slot = low.key_slot
    blob.group_cap, blob.merge_cap = 4, 16
    blob.prog = low.program.to_struct()
    blob.spec = low.spec()
    blob.fin = fin
    if fin_prog is not None:
        blob.fin_prog = fin_prog
    for o, (name, ctype) in enumerate(out_schema):
        blob.out_types[o] = _TYPE_CODE[ctype]
        blob.out_names[o].value = name[len(prefix):].encode()[:63] if prefix and name.startswith(prefix) else name.encode()[:63]
    return blob, Path(scan.producer.file_path), out_schema


def lower_join_stage_plan(full_task: Any, plan: Any = None, n_parts: int | None = None) -> tuple[hs.hs_join_stage_plan, Path, Path, Schema]:
    """orders JOIN lineitem ... GROUP BY (BASELINE config 4's shape) -> (plan blob of the native JOIN stage, build table path,
    probe table path, result schema).  The reference plans such a query as four stages: shuffle of
    either input by the join key, [join -> partial aggregate -> shuffle], final; the native stage runs the last two
    over the tables themselves, with the JoinJob of a row given by hash(key) % n_parts (plan.py:99-109, tasks.py:362)."""
    from . import constants  # noqa: PLC0415

    if plan is None:
        from .plan import PhysicalPlan  # noqa: PLC0415

        plan = PhysicalPlan.generate_physical_plan(full_task)
    stages = list(plan.stages)
    join = next((st for st in stages if _cls(st.producer) == "BroadcastHashJoinTask"), None)
    final = next((st for st in stages if _cls(st.producer) == "LoadShuffleFilesTask" and _cls(st.writer) == "WriteToLocalFileTask"), None)
    if join is None or final is None or len(stages) != 4 or len(join.dependencies) != 2:
        raise StageUnsupported("not a [scan, scan, join -> partial aggregate, final] plan")
    sides = []
    for dep in join.dependencies:  # the two inputs: plain scans, at most a projection that only selects columns
        if _cls(dep.producer) != "LoadTableBlockTask":
            raise StageUnsupported("a join input is not a table scan")
        for task in dep.consumers:
            if _cls(task) != "ProjectTask" or any(_cls(_bare(c)) not in ("Col", "SchemaCol") for c in task.columns):
                raise StageUnsupported(f"{_cls(task)} between a table and the join")
        sides.append(dep.producer)
    build, probe = sides
    task = join.producer
    filters, partial = [], None
    for t in join.consumers:
        if _cls(t) == "FilterTask" and partial is None:
            filters.append(t.condition)
        elif _cls(t) == "AggregateTask" and t.before_shuffle and partial is None:
            partial = t
        else:
            raise StageUnsupported(f"{_cls(t)} in the join stage")
    consumers = list(final.consumers)
    if partial is None or not consumers or _cls(consumers[0]) != "AggregateTask" or consumers[0].before_shuffle:
        raise StageUnsupported("no partial / final aggregate pair")
    if len(consumers) > 2 or (len(consumers) == 2 and _cls(consumers[1]) != "ProjectTask"):
        raise StageUnsupported("more than a projection after the final aggregate")
    merge = consumers[0]
    project = list(consumers[1].columns) if len(consumers) == 2 else None
    out_schema = list(final.writer.inferred_schema)

    def table_names(producer: Any) -> list[str]:
        prefix = f"{producer.alias}." if getattr(producer, "alias", "") else ""
        return [prefix + n for n, _ in BlockFile(Path(producer.file_path)).file_schema]

    bnames, pnames = table_names(build), table_names(probe)
    bschema, pschema = list(BlockFile(Path(build.file_path)).file_schema), list(BlockFile(Path(probe.file_path)).file_schema)
    lname, rname = task.left_key.name, task.right_key.name
    if lname not in bnames or rname not in pnames:
        raise StageUnsupported("join keys are not plain columns of the two tables")
    # the aggregate's view: every probe-side column under its name + the build-side columns it names
    used = set()
    for expr in [*filters, partial.group_by_column, *[a.original_col for a in partial.agg_columns]]:
        used.update(c.name for c in expr.all_nested_columns if _cls(c) in ("Col", "SchemaCol"))
    wanted_build = sorted(n for n in used if n in bnames and n not in pnames)
    if len(wanted_build) > 1:
        raise StageUnsupported("the aggregate reads more than one build-side column")
    payload = wanted_build[0] if wanted_build else None
    if payload is not None and any(c.name == payload for f in filters for c in f.all_nested_columns if _cls(c) in ("Col", "SchemaCol")):
        raise StageUnsupported("a predicate on the build-side column")  # it exists as a table byte only
    schema = [(n, t) for n, (_, t) in zip(pnames, pschema)]
    if payload is not None:
        if bschema[bnames.index(payload)][1] != ColumnType.STRING:
            raise StageUnsupported("the build-side column must be a STRING column")
        schema.append((payload, ColumnType.STRING))
    kinds = [_FILE_KIND[t] for _, t in schema]
    # the payload is lowered as a dictionary-coded string (one code byte): the dictionary itself is built natively, its
    # contents do not matter to the program as long as no predicate looks inside the strings
    dicts = [None] * len(pnames) + ([(b"",)] if payload is not None else [])
    low = lower_aggregate(schema, kinds, filters, partial.group_by_column, partial.agg_columns, dicts)
    if low.program.code_columns:
        raise StageUnsupported("a predicate on dictionary codes")
    if low.numeric_slots >= hs.HS_FUSED_COLS:
        raise StageUnsupported(f"more than {hs.HS_FUSED_COLS - 1} column slots")
    acc_kinds = [hs.I32 if is_int else hs.F32 for is_int in low.acc_is_int]
    key_idx = low.program.columns[low.key_slot]
    fin, fin_prog, outs = lower_finish(low.agg_to_acc, acc_kinds, kinds[key_idx], merge.agg_columns, merge.inferred_schema,
                                       project, out_schema)
    blob = hs.hs_join_stage_plan()
    blob.version = hs.HS_JOIN_STAGE_PLAN_VERSION
    blob.build_key_col = bnames.index(lname)
    blob.build_payload_col = bnames.index(payload) if payload is not None else -1
    blob.probe_key_col = pnames.index(rname)
    blob.n_parts = n_parts if n_parts is not None else constants.SHUFFLE_PARTITIONS
    blob.n_cols = len(low.program.columns)
    for slot, idx in enumerate(low.program.columns):
        blob.col_ids[slot] = idx if idx < len(pnames) else -1
    blob.key_slot = low.key_slot
    blob.group_cap, blob.merge_cap = 4, 16
    blob.prog = low.program.to_struct()
    blob.spec = low.spec()
    blob.fin = fin
    if fin_prog is not None:
        blob.fin_prog = fin_prog
    for o, (name, ctype) in enumerate(out_schema):
        blob.out_types[o] = _TYPE_CODE[ctype]
        blob.out_names[o].value = name.split(".", 1)[-1].encode()[:63] if "." in name else name.encode()[:63]
    return blob, Path(build.file_path), Path(probe.file_path), out_schema


def lower_select_stage_plan(full_task: Any, plan: Any = None) -> tuple[hs.hs_select_stage_plan, Path, Schema]:
    """table -> [filter]* -> [select] (one stage, rows to the result file) -> (plan blob of the native SELECT / WHERE stage,
    table path, result schema)."""
    from .lowering import ProgramBuilder, unalias  # noqa: PLC0415

    if plan is None:
        from .plan import PhysicalPlan  # noqa: PLC0415

        plan = PhysicalPlan.generate_physical_plan(full_task)
    stages = list(plan.stages)
    if len(stages) != 1 or _cls(stages[0].producer) != "LoadTableBlockTask" or _cls(stages[0].writer) != "WriteToLocalFileTask":
        raise StageUnsupported("not a one-stage scan to the result file")
    stage = stages[0]
    filters, project = [], None
    for task in stage.consumers:
        if _cls(task) == "FilterTask" and project is None:
            filters.append(task.condition)
        elif _cls(task) == "ProjectTask" and project is None:
            project = task
        else:
            raise StageUnsupported(f"{_cls(task)} after the projection")
    out_schema = list(stage.writer.inferred_schema)
    table_schema = list(stage.producer.inferred_schema)
    names = [n for n, _ in table_schema]
    kinds = [_FILE_KIND[t] for _, t in table_schema]
    blob = hs.hs_select_stage_plan()
    blob.version = hs.HS_SELECT_STAGE_PLAN_VERSION
    if filters:
        cond = filters[0]
        for extra in filters[1:]:
            cond = cond & extra
        fb = ProgramBuilder(table_schema, kinds)
        if fb.emit_out(0, cond) != "B":
            fb = ProgramBuilder(table_schema, kinds)
            fb.emit_out(0, cond != 0)
        fprog = fb.finish()
        blob.n_cols = len(fprog.columns)
        for slot, idx in enumerate(fprog.columns):
            blob.col_ids[slot] = idx
        blob.filter = fprog.to_struct()
    columns = list(project.columns) if project is not None else None
    if columns is None:  # every column as it is
        if len(out_schema) != len(table_schema):
            raise StageUnsupported("writer schema differs from the table's")
        srcs = list(range(len(table_schema)))
    else:
        pb = ProgramBuilder(table_schema, kinds)
        srcs, n_prog = [], 0
        for o, col in enumerate(columns):
            bare = unalias(col)
            if _cls(bare) in ("Col", "SchemaCol"):
                if bare.name not in names:
                    raise ValueError(f'Column "{bare.name}" not found in schema {table_schema}')
                srcs.append(names.index(bare.name))
                continue
            if pb.string_tag(bare):
                raise StageUnsupported("string expression in the projection")
            if n_prog >= hs.HS_MAX_OUTS:
                raise StageUnsupported("too many computed columns")
            tag = pb.emit_out(n_prog, col)
            if tag == "B":
                raise AssertionError("a comparison cannot be selected as a column (the reference has no BOOL type)")
            want = out_schema[o][1]
            if (tag, want) not in (("F", ColumnType.FLOAT), ("I", ColumnType.INTEGER)):
                raise StageUnsupported(f"computed column of tag {tag} stored as {want}")
            blob.project_kinds[n_prog] = hs.F64 if tag == "F" else hs.I64
            srcs.append(-1 - n_prog)
            n_prog += 1
        if n_prog:
            pprog = pb.finish()
            blob.n_pcols = len(pprog.columns)
            for slot, idx in enumerate(pprog.columns):
                blob.pcol_ids[slot] = idx
            blob.project = pprog.to_struct()
    if len(srcs) != len(out_schema) or len(srcs) > hs.HS_FINISH_MAX_OUT:
        raise StageUnsupported("result schema does not match the selected columns")
    blob.n_out = len(srcs)
    prefix = f"{stage.producer.alias}." if getattr(stage.producer, "alias", "") else ""
    for o, ((name, ctype), src) in enumerate(zip(out_schema, srcs)):
        blob.out_src[o] = src
        blob.out_types[o] = _TYPE_CODE[ctype]
        blob.out_names[o].value = name[len(prefix):].encode()[:63] if prefix and name.startswith(prefix) else name.encode()[:63]
    return blob, Path(stage.producer.file_path), out_schema


class NativeSelectStage:
    """A prepared select / where query behind the C ABI: ``run(path)`` -> rows (through the result BlockFile the library writes)."""

    def __init__(self, engine: "NativeEngine", full_task: Any, plan: Any = None) -> None:
        self.engine, self.lib = engine, engine.lib
        self.blob, self.table_path, self.schema = lower_select_stage_plan(full_task, plan)
        self.handle = C.c_void_p()
        hs.check(self.lib.hs_select_stage_prepare(engine.handle, engine.table(self.table_path), C.byref(self.blob),
                                                  C.sizeof(self.blob), C.byref(self.handle)), "hs_select_stage_prepare")

    def run(self, out_path: Path | str, rows_per_block: int | None = None, stream: int | None = None) -> list[Row]:
        from . import constants  # noqa: PLC0415

        flags, nrows = C.c_uint32(0), C.c_int64(0)
        hs.check(self.lib.hs_select_stage_run(self.handle, stream, C.byref(flags), C.byref(nrows)), "hs_select_stage_run")
        raise_for_flags(flags.value)
        if nrows.value == 0:
            return []
        Path(out_path).parent.mkdir(parents=True, exist_ok=True)
        hs.check(self.lib.hs_select_result_write_blockfile(self.handle, str(out_path).encode(),
                                                           rows_per_block or constants.ROWS_PER_BLOCK), "hs_select_result_write_blockfile")
        return read_result_file(out_path)

    def close(self) -> None:
        if self.handle:
            self.lib.hs_select_stage_destroy(self.handle)
            self.handle = C.c_void_p()


def _bare(col: Any) -> Any:
    from .lowering import unalias  # noqa: PLC0415

    return unalias(col)


class NativeJoinStage:
    """A prepared join + GROUP BY query behind the C ABI: ``run()`` -> rows (through the result BlockFile the library writes)."""

    def __init__(self, engine: "NativeEngine", full_task: Any, plan: Any = None) -> None:
        self.engine, self.lib = engine, engine.lib
        self.blob, self.build_path, self.probe_path, self.schema = lower_join_stage_plan(full_task, plan)
        self.handle = C.c_void_p()
        hs.check(self.lib.hs_join_stage_prepare(engine.handle, engine.table(self.build_path), engine.table(self.probe_path),
                                                C.byref(self.blob), C.sizeof(self.blob), C.byref(self.handle)), "hs_join_stage_prepare")

    def run(self, out_path: Path | str, stream: int | None = None) -> list[Row]:
        flags, nrows = C.c_uint32(0), C.c_int64(0)
        hs.check(self.lib.hs_join_stage_run(self.handle, stream, C.byref(flags), C.byref(nrows)), "hs_join_stage_run")
        raise_for_flags(flags.value)
        if nrows.value == 0:
            return []
        Path(out_path).parent.mkdir(parents=True, exist_ok=True)
        hs.check(self.lib.hs_join_result_write_blockfile(self.handle, str(out_path).encode()), "hs_join_result_write_blockfile")
        return read_result_file(out_path)

    def stats(self) -> dict:
        s = (C.c_int64 * 8)()
        hs.check(self.lib.hs_join_stage_stats(self.handle, s), "hs_join_stage_stats")
        return dict(zip(("runs", "replays", "grows", "group_cap", "merge_cap", "dictionary", "table_slots", "unit_cap"), (int(v) for v in s)))

    def close(self) -> None:
        if self.handle:
            self.lib.hs_join_stage_destroy(self.handle)
            self.handle = C.c_void_p()


class NativeEngine:
    """hs_engine + the tables it has open."""

    def __init__(self, device: int = 0) -> None:
        self.lib = hs.load_library()
        self.handle = C.c_void_p()
        hs.check(self.lib.hs_engine_create(device, C.byref(self.handle)), "hs_engine_create")
        self._tables: dict[tuple, C.c_void_p] = {}

    def table(self, path: Path | str, rank: int = 0, world: int = 1) -> C.c_void_p:
        key = (str(Path(path).resolve()), rank, world)
        if key not in self._tables:
            t = C.c_void_p()
            hs.check(self.lib.hs_table_open(self.handle, key[0].encode(), rank, world, C.byref(t)), "hs_table_open")
            self._tables[key] = t
        return self._tables[key]

    def close(self) -> None:
        for t in self._tables.values():
            self.lib.hs_table_close(t)
        self._tables.clear()
        if self.handle:
            self.lib.hs_engine_destroy(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self) -> "NativeEngine":
        return self

    def __exit__(self, *exc: Any) -> None:
        self.close()


class NativeStage:
    """A prepared query: ``run()`` -> result rows; ``write(path)`` -> the result BlockFile."""

    def __init__(self, engine: NativeEngine, full_task: Any, plan: Any = None, world: int = 1, rank: int = 0) -> None:
        self.engine, self.lib = engine, engine.lib
        self.blob, self.table_path, self.schema = lower_stage_plan(full_task, plan)
        self.handle = C.c_void_p()
        hs.check(self.lib.hs_stage_prepare(engine.handle, engine.table(self.table_path, rank, world), C.byref(self.blob),
                                           C.sizeof(self.blob), world, C.byref(self.handle)), "hs_stage_prepare")

    def run(self, stream: int | None = None) -> list[Row]:
        flags, nrows = C.c_uint32(0), C.c_int64(0)
        hs.check(self.lib.hs_stage_run(self.handle, stream, C.byref(flags), C.byref(nrows)), "hs_stage_run")
        raise_for_flags(flags.value)
        return list(rows_from_raw(self.schema, self.raw_columns()))

    def raw_columns(self) -> list[Any]:
        cols = (hs.hs_result_col * hs.HS_FINISH_MAX_OUT)()
        n = C.c_int32(0)
        hs.check(self.lib.hs_result_columns(self.handle, cols, hs.HS_FINISH_MAX_OUT, C.byref(n)), "hs_result_columns")
        raw: list[Any] = []
        for o in range(n.value):
            c = cols[o]
            rows, nbytes = int(c.n_rows), int(c.n_rows) * int(c.width)
            buf = np.frombuffer((C.c_uint8 * nbytes).from_address(c.data), dtype=np.uint8).copy() if nbytes else np.zeros(0, np.uint8)
            raw.append(StrCol(np.full(rows, c.width, np.uint8), buf) if c.kind == hs.STR else buf.view(_NP[c.kind]))
        return raw

    def write(self, path: Path | str) -> Path:
        Path(path).parent.mkdir(parents=True, exist_ok=True)
        hs.check(self.lib.hs_result_write_blockfile(self.handle, str(path).encode()), "hs_result_write_blockfile")
        return Path(path)

    def stats(self) -> dict:
        s = (C.c_int64 * 6)()
        hs.check(self.lib.hs_stage_stats(self.handle, s), "hs_stage_stats")
        return dict(zip(("runs", "replays", "grows", "group_cap", "merge_cap", "chunks"), (int(v) for v in s)))

    def close(self) -> None:
        if self.handle:
            self.lib.hs_stage_destroy(self.handle)
            self.handle = C.c_void_p()


def raise_for_flags(flags: int) -> None:
    """Data-dependent failures surface as the exceptions the reference's Python raises (same table as Device)."""
    if flags & hs.FLAG_DIV_ZERO:
        raise ZeroDivisionError("division by zero")
    if flags & hs.FLAG_INT_OVERFLOW:
        raise OverflowError("int too big to convert")
    if flags & hs.FLAG_FLT_OVERFLOW:
        raise OverflowError("float too large to pack with f format")
    if flags & hs.FLAG_TYPE_ASSERT:
        raise AssertionError("FLOAT column holds int")
    if flags & hs.FLAG_BAD_PROGRAM:
        raise RuntimeError("internal error: device interpreter rejected the program")


def read_result_file(path: Path | str) -> list[Row]:
    return list(BlockFile(Path(path)).read_data_rows())
