"""Engine plug-in surface + the HIP engine (reference: src/mini_spark/execution.py:35-93).

``ExecutionEngine`` mirrors the reference's abstract base: ``execute_full_task(Task) -> list[JobResult]``
is the one method an engine must provide; ``collect_results`` reads the returned ``OutputFile``s back as
row dicts.  ``HipExecutionEngine`` is the drop-in: it accepts a task chain built either by
:mod:`minispark_amd.dataframe` or by the reference's own ``DataFrame`` (tasks are inspected by class and
attribute *names*), plans it with the matching planner, and runs every stage on one MI355X through
libhipspark.so.  Results are written as BlockFiles it owns and deletes in ``__exit__``.

Stage execution keeps the reference's semantics-bearing structure:

* scan stage: one partial-aggregation unit per file block; partials are quantised to f32/i32 exactly
  where the reference writes its shuffle file;
* join stage: one unit per shuffle partition ``hash(key) % 10``;
* final stage: partial rows of a key are merged in unit order in fp64, AVG is ``sum / count`` on the
  un-rounded merged sum, the result file rounds to f32 again.

Inter-stage "shuffle files" stay in HBM as device batches (already quantised to the file types).
"""

from __future__ import annotations

import math
import os
import sys
import shutil
import uuid
from abc import ABC, abstractmethod
from contextlib import AbstractContextManager
from pathlib import Path
from typing import Any, Iterator, Sequence

from . import constants
from .constants import Row, Schema
from .io import BlockFile
from .jobs import JobResult, ResultFile


class ExecutionError(Exception):
    def __init__(self, message: str = "Execution failed") -> None:
        super().__init__(message)


PRIVATE_TIER_MAX = 16    # dictionary slots per workgroup up to which every lane keeps private accumulators
SHARED_TIER_MAX = 4096   # ... and up to which one LDS dictionary per workgroup is used; beyond: the HBM tier


class RestartQuery(Exception):
    """Run the query again from the start (a fused path found at run time that it does not apply)."""


class ExecutionEngine(AbstractContextManager, ABC):
    @abstractmethod
    def execute_full_task(self, full_task: Any) -> list[JobResult]: ...

    def generate_physical_plan(self, full_task: Any) -> Any:
        # a task chain built by the reference is planned by the reference's planner (drop-in use);
        # otherwise by this package's counterpart
        if type(full_task).__module__.startswith("mini_spark."):
            from mini_spark.plan import PhysicalPlan as ReferencePlan  # type: ignore[import-not-found]  # noqa: PLC0415

            return ReferencePlan.generate_physical_plan(full_task)
        from .plan import PhysicalPlan  # noqa: PLC0415

        return PhysicalPlan.generate_physical_plan(full_task)

    def collect_results(self, results: list[JobResult], limit: float = math.inf) -> Iterator[Row]:
        output_files = {file for result in results for file in result.output_files}
        if limit == math.inf and len(output_files) == 1:
            (file,) = output_files
            if isinstance(file, ResultFile):  # rows already in host memory: hand the list over, no generator per row
                return iter(file.rows())
        return self._collect_results(output_files, limit)

    def collect_columns(self, results: list[JobResult]) -> dict:
        """The result column-wise (name -> numpy array / list of str) - no row dicts are built.  An addition to the
        reference's surface for large result sets: 125 000 groups cost 40 ms as dicts, none as columns."""
        output_files = [file for result in results for file in result.output_files]
        if not output_files:
            return {}
        if len(output_files) == 1 and isinstance(output_files[0], ResultFile):
            return output_files[0].columns()
        import numpy as np  # noqa: PLC0415

        from .io import StrCol  # noqa: PLC0415

        merged: dict = {}
        for file in output_files:
            bf = BlockFile(file.file_path)
            for block_id in range(len(bf.block_starts)):
                for (name, _), raw in zip(bf.file_schema, bf.read_block_raw(block_id)):
                    merged.setdefault(name, []).append(raw.to_list() if isinstance(raw, StrCol) else raw)
        return {name: (np.concatenate(parts) if isinstance(parts[0], np.ndarray) else [v for p in parts for v in p])
                for name, parts in merged.items()}

    def _collect_results(self, output_files: Any, limit: float) -> Iterator[Row]:
        for file in output_files:
            rows = file.rows() if isinstance(file, ResultFile) else BlockFile(file.file_path).read_data_rows()
            if limit == math.inf:
                yield from rows
                continue
            for row in rows:
                yield row
                limit -= 1
                if limit <= 0:
                    return

    def sql(self, query: str) -> Any:
        """SQL text -> DataFrame bound to this engine (reference execution.py:57-62).  Parsed by this package's
        own parser of the reference's grammar (minispark_amd/parser.py): no third-party dependency."""
        from .parser import parse_sql  # noqa: PLC0415

        return parse_sql(query, self)


_UID = iter(range(1, 1 << 62))


def _uid(obj: Any) -> int:
    """Identity token for cache keys.  id() is NOT one: plans, tasks and tables are freed when they fall out of the
    engine's caches and a later object can get the same id - together with recycled device addresses that made a
    recorded run of one query answer another (fuzz seed 467).  The token lives on the object and is never reused."""
    token = getattr(obj, "_hs_uid", None)
    if token is None:
        token = next(_UID)
        try:
            obj._hs_uid = token
        except AttributeError:  # an object without a __dict__: fall back to its id (kept alive by its cache entry)
            return id(obj)
    return token


def _cls(obj: Any) -> str:
    return type(obj).__name__


def _plain_names(expr: Any) -> list[str]:
    return [c.name for c in expr.all_nested_columns if _cls(c) in ("Col", "SchemaCol")]


class _DeferredScan:
    """Output of a scan stage that was NOT run: the probe side of a join, larger than the HBM budget.  The join stage
    reads it block range by block range (HipExecutionEngine._run_join_stage_streamed)."""

    def __init__(self, stage: Any, ranges: list[list[int]]) -> None:
        self.stage, self.ranges = stage, ranges


class HipExecutionEngine(ExecutionEngine):
    """Runs query stages on an MI355X.  Zero-argument constructible like the reference's engines."""

    def __init__(self, device: int | None = None, work_folder: Path | None = None,
                 trace_file: str | Path | None = None) -> None:
        from .device import Device  # noqa: PLC0415 - loads libhipspark.so and needs a GPU: fail loudly here

        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self.dev = Device(device)
        self._executor_id = f"hip:{self.dev.index}"
        self._work_folder = work_folder
        self._tables: dict[str, Any] = {}
        self._plans: dict[Any, Any] = {}
        self._recordings: dict[Any, Any] = {}  # plan key -> device.Recording of a fully device-resident run
        self.replay_enabled = os.environ.get("HIPSPARK_REPLAY", "1") != "0"
        self.replays = 0
        self.short_tail_enabled = os.environ.get("HIPSPARK_SHORT_TAIL", "1") != "0"
        self.shared_tier_enabled = os.environ.get("HIPSPARK_SHARED_TIER", "1") != "0"
        # round 2: dictionary-coded string columns and the in-place unique-key join (4.6)
        self.dict_enabled = os.environ.get("HIPSPARK_DICT", "1") != "0"
        self.fused_join_enabled = os.environ.get("HIPSPARK_FUSED_JOIN", "1") != "0"
        self.fused_probe_enabled = os.environ.get("HIPSPARK_FUSED_PROBE", "1") != "0"  # round 3: probe inside the aggregate
        self.fused_probes = 0
        # N ranks, round 4: the join's byte table built from ROUTED build rows, only where the rank's own probe blocks reach
        # (0: every rank all-gathers the build side and builds the whole table - also the form for unclustered probe tables)
        self.sharded_build_enabled = os.environ.get("HIPSPARK_SHARDED_BUILD", "1") != "0"
        self.sharded_builds = 0
        self._no_join8: set[Any] = set()  # join task ids the fused probe turned out not to hold
        self._no_fused_join: set[Any] = set()  # join task ids whose build side turned out to hold duplicate keys
        self._fused_join_tasks: set[Any] = set()  # join task ids the running query took the in-place path for
        self._lds_merges: list[int] = []  # final-merge tasks the running query folded on chip (see HS_FLAG_MERGE_ROWS)
        budget = os.environ.get("HIPSPARK_HBM_BUDGET")
        self.hbm_budget: int | None = int(float(budget)) if budget else None  # bytes of referenced columns kept resident
        self.streamed_ranges = 0
        self._version = 0  # bumped by everything that could invalidate a validated recording (see _execute_full_task)
        self.fused_joins = 0
        self._no_short_tail: set[Any] = set()  # partial AggregateTask ids that must take the general path
        self.short_tails = 0  # queries finished by the short tail (first runs and recordings; replays count in `replays`)
        self._p2p: Any = None  # peer-to-peer slab exchange (prototype, HIPSPARK_P2P_SLABS=1), set up at its first use
        self.p2p_exchanges = 0
        self._plan_runs: dict[Any, int] = {}
        self._owned_dirs: set[Path] = set()
        self._result_root: Path | None = None
        self._result_paths: dict[Any, Path] = {}
        self._job_prefix, self._job_seq = uuid.uuid4().hex[:12], 0
        # dictionary capacities (partial aggregate per workgroup / final merge) grow when a run overflows them; they
        # are remembered per QUERY SHAPE, so a query with many groups does not slow down the next one with few
        self._distinct_accs: dict[int, int] = {}  # partial AggregateTask id -> distinct accumulators after lowering
        self._caps_by_shape: dict[Any, dict] = {}
        self._caps: dict = {"group": 4, "merge": 16, "merge_overflowed": False}
        self._global_partial: set[int] = set()  # AggregateTask ids (of cached plans) running on the global tier
        self._global_merge: set[int] = set()
        self.dist: Any = None  # torch.distributed once enable_distributed() was called
        self.rank, self.world = 0, 1
        self._remote_flags: Any = None
        # tracing (reference utils.py:83-135): spans per query / stage / replay + the scan kernel on a GPU track
        self.trace_file = trace_file or os.environ.get("HIPSPARK_TRACE") or None
        self.tracer: Any = None
        if self.trace_file:
            from .tracing import Tracer  # noqa: PLC0415

            self.tracer = Tracer()
            self._gpu_track = self.tracer.new_track(f"GPU {self.dev.index}: every launch (hs_trace)")
            self._scan_track = self.tracer.new_track(f"GPU {self.dev.index}: scan kernel (event pair)")
            self.dev.time_scan_kernel(True)

    @property
    def group_cap_hint(self) -> int:
        return self._caps["group"]

    @group_cap_hint.setter
    def group_cap_hint(self, value: int) -> None:
        self._caps["group"] = value

    @property
    def merge_cap_hint(self) -> int:
        return self._caps["merge"]

    @merge_cap_hint.setter
    def merge_cap_hint(self, value: int) -> None:
        self._caps["merge"] = value

    @property
    def _merge_overflowed(self) -> bool:
        return self._caps["merge_overflowed"]

    @_merge_overflowed.setter
    def _merge_overflowed(self, value: bool) -> None:
        self._caps["merge_overflowed"] = value

    def _private_tier_fits(self, task: Any, batch: Any, pending: Sequence[Any]) -> bool:
        """Per-lane private accumulator tables pay while a 256-lane workgroup's tables leave room for a second
        workgroup on the CU: group_cap x DISTINCT accumulators x 256 lanes x 8 bytes <= 64 KiB (Q1: 4 x 6 -> 48 KiB;
        its 11 carried aggregates share 6 accumulators).  Beyond that the shared LDS dictionary is faster
        (measured: 11 groups x 2 aggregates 0.90 ms private vs 0.25 ms shared)."""
        cap = self.group_cap_hint
        if cap > PRIVATE_TIER_MAX:
            return False
        n_acc = self._distinct_accs.get(_uid(task))
        if n_acc is None:
            from .lowering import lower_aggregate  # noqa: PLC0415

            try:
                n_acc = len(lower_aggregate(batch.schema, batch.kinds, pending, task.group_by_column, task.agg_columns,
                                            batch.dicts).acc_ops)
            except Exception:  # noqa: BLE001 - whatever lowering objects to is reported by the tier that runs it
                n_acc = len(task.agg_columns)
            self._distinct_accs[_uid(task)] = n_acc
        return cap * 16 + cap * max(n_acc, 1) * 256 * 8 <= 64 * 1024

    def _select_caps(self, plan: Any) -> None:
        """Point the capacity hints at the entry of this plan's query shape (tasks, expressions, table paths)."""
        shape = getattr(plan, "_hs_shape", None)
        if shape is None:
            parts = []
            for stage in plan.stages:
                for task in (stage.producer, *stage.consumers, stage.writer):
                    attrs = [str(getattr(task, name)) for name in ("file_path", "alias", "condition", "columns",
                                                                   "group_by_column", "agg_columns", "join_condition",
                                                                   "key_column", "before_shuffle") if hasattr(task, name)]
                    parts.append((_cls(task), *attrs))
            shape = plan._hs_shape = tuple(parts)
        caps = self._caps_by_shape.get(shape)
        if caps is None:
            if len(self._caps_by_shape) >= 256:
                self._caps_by_shape.pop(next(iter(self._caps_by_shape)))
            caps = self._caps_by_shape[shape] = {"group": 4, "merge": 16, "merge_overflowed": False}
        self._caps = caps

    # ---- context manager -------------------------------------------------------------------------------
    def __exit__(self, exc_type, exc_value, traceback) -> None:  # noqa: ANN001
        if self.tracer is not None:
            self.tracer.save(self.trace_file)
        if self._p2p:
            import torch  # noqa: PLC0415

            torch.cuda.synchronize(self.dev.device)  # no launch of ours still reads the mapped buffers
            if self.dist is not None and exc_type is None:
                self.dist.barrier(group=self.group)  # ... and no peer still writes into ours
            self._p2p.close()
            self._p2p = None
        for d in self._owned_dirs:
            shutil.rmtree(d, ignore_errors=True)
        self._owned_dirs.clear()
        self._result_paths.clear()
        self._result_root = None

    # ---- multi-GPU -------------------------------------------------------------------------------------
    def enable_distributed(self, dist: Any, group: Any = None) -> None:
        """One process per GPU: this engine owns the file blocks b with b % world == rank; partial
        aggregates meet through one all-gather per query (minispark_amd/distributed.py); rank 0 writes
        the result, the other ranks return no output files."""
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._version += 1

    # ---- tables --------------------------------------------------------------------------------------
    def attach_device_table(self, path: str | Path, table: Any) -> None:
        """Register columns that already live in HBM (synthetic data) as the table stored at ``path``.
        ``path`` must hold a BlockFile header with the same schema (its blocks are not read)."""
        self._tables[str(Path(path).resolve())] = table
        self._version += 1

    def _table(self, path: Path) -> Any:
        from . import table as tbl  # noqa: PLC0415

        key = str(Path(path).resolve())
        cached = self._tables.get(key)
        if cached is not None and (cached.stamp == () or cached.stamp == tbl.file_stamp(Path(path))):
            return cached
        opened = tbl.open_table(Path(path), self.rank, self.world, distributed=self.dist is not None)
        self._tables[key] = opened
        return opened

    # ---- the plug-in entry point ---------------------------------------------------------------------------
    def execute_full_task(self, full_task: Any) -> list[JobResult]:
        if self.tracer is None:
            return self._execute_full_task(full_task)
        import time  # noqa: PLC0415

        self.tracer.start("query")
        self.dev.last_scan = None  # set again by a partial aggregate of this query (or by its replayed recording)
        lib, stream = self.dev._raw_lib, self.dev.stream
        t_base = time.time_ns()
        tracing_gpu = lib.hs_trace_begin(stream) == 0
        try:
            return self._execute_full_task(full_task)
        finally:
            self.tracer.end()
            if tracing_gpu:
                # every launch of the query - scan, unit combine, finish, gathers, join build / probe ... - as a slice
                # on the GPU track, placed at (host time of the query start + its offset on the stream)
                import ctypes as C  # noqa: PLC0415

                from . import hipspark as hs  # noqa: PLC0415

                slices = (hs.hs_trace_slice * 512)()
                n = C.c_int32(0)
                if lib.hs_trace_end(stream, slices, 512, C.byref(n)) == 0:
                    for i in range(n.value):
                        sl = slices[i]
                        self.tracer.complete(sl.name.decode(errors="replace"), t_base + int(sl.start_us * 1e3),
                                             int(sl.dur_us * 1e3), self._gpu_track)
            if getattr(self.dev, "last_scan", None) is not None and self.dev.scan_events is not None:
                try:  # the dominant kernel once more with its launch geometry as arguments (bench.py's event pair)
                    ms = self.dev.scan_kernel_ms()
                    self.tracer.complete("scan kernel (k_agg_jit / k_agg_main)", time.time_ns() - int(ms * 1e6),
                                         int(ms * 1e6), self._scan_track, **self.dev.last_scan)
                except RuntimeError:
                    pass  # no aggregate ran in this query: the events were never recorded

    def _execute_full_task(self, full_task: Any) -> list[JobResult]:
        from .device import DeviceError  # noqa: PLC0415
        from .hipspark import HipSparkError  # noqa: PLC0415

        from .device import RetryWithLargerDictionary  # noqa: PLC0415

        # hot loop: the same task object as last time, nothing about the engine's tables / capacities / recordings has
        # changed since its recording was last validated (self._version) and its table files are untouched
        fast = getattr(full_task, "_hs_fast", None)
        if fast is not None and fast[0] is self and fast[1] == self._version and self.tracer is None:
            if fast[4] == self._switches() and all(self._stamp_of(path) == stamp for path, stamp in fast[3]):
                replayed = self._replay(fast[2])
                if replayed is not None:
                    return replayed
            full_task._hs_fast = None
        for _attempt in range(12):
            plan = self._cached_plan(full_task)
            self._select_caps(plan)
            rec_key = self._recording_key(plan)
            rec = self._recordings.get(rec_key) if self.replay_enabled else None
            if rec is not None:
                if self.tracer is not None:
                    self.tracer.start("replay of the recorded launch sequence")
                replayed = self._replay(rec)
                if self.tracer is not None:
                    self.tracer.end()
                if replayed is not None:
                    try:
                        full_task._hs_fast = (self, self._version, rec, self._file_stamps(plan), self._switches())
                    except AttributeError:
                        pass
                    return replayed
                del self._recordings[rec_key]  # something data-dependent changed: take the full path again
            self._version += 1  # a full run may load / re-code tables, grow capacities, replace recordings
            self.dev.reset_flags()
            self._fused_join_tasks.clear()
            self._lds_merges: list[int] = []
            outputs: dict[int, Any] = {}
            results: list[JobResult] = []
            # scan stages that feed a join's probe (right) side: one that does not fit HBM is not materialised - the join
            # stage streams it block range by block range (the reference's right side is streamed too: tasks.py:224-240)
            self._probe_stages = {id(st.dependencies[1]) for st in plan.stages
                                  if _cls(st.producer) == "BroadcastHashJoinTask" and len(st.dependencies) == 2}
            # record the second (cache-warm) run of a plan: by then every buffer it needs is prepared
            want_record = self.replay_enabled and self._plan_runs.get(rec_key, 0) >= 1
            recording = self.dev.start_recording() if want_record else None
            try:
                for stage in plan.stages:
                    if self.tracer is not None:
                        self.tracer.start(f"stage {stage.stage_id}: {_cls(stage.producer)} -> {_cls(stage.writer)}")
                    try:
                        results = self._run_stage(stage, outputs)
                    finally:
                        if self.tracer is not None:
                            self.tracer.end()
                    stage.job_results.extend(results)
                # counted under the key the NEXT run will compute: a first run may re-code table columns (dictionaries)
                # or learn capacities, which are part of the key
                done_key = self._recording_key(plan)
                self._plan_runs[done_key] = self._plan_runs.get(done_key, 0) + 1
                if recording is not None:
                    self.dev.stop_recording()
                    recording.scan_info = getattr(self.dev, "last_scan", None)
                    if not recording.poisoned and recording.finish is not None and recording.result is not None:
                        if len(self._recordings) >= 8:
                            self._recordings.pop(next(iter(self._recordings)))
                        self._recordings[rec_key] = recording
                return results
            except RestartQuery:
                self.dev.stop_recording()
            except RetryWithLargerDictionary as grow:
                self.dev.stop_recording()
                # more distinct GROUP BY keys than a dictionary was sized for: grow and re-run; past the on-chip
                # limits the stages switch to the global-memory tier (TierExceeded in _run_stage).  The per-unit
                # capacity (x2 while the private-table tier applies, x4 beyond; every step re-runs the scan) and the
                # final merge's (x4, on-chip up to 4096 slots) grow on their own flags: a query with 50 groups in 29
                # units climbs 4 -> 64 per unit and must not end up with an "overflowed" merge of 1 450 rows
                if grow.unit_full:
                    self.group_cap_hint *= 2 if self.group_cap_hint < PRIVATE_TIER_MAX else 4
                if grow.merge_full:
                    if self.merge_cap_hint >= 4096:
                        self._merge_overflowed = True
                    self.merge_cap_hint = min(self.merge_cap_hint * 4, 4096)
                # the merge sees at least the keys of one unit
                self.merge_cap_hint = min(max(self.merge_cap_hint, 4 * self.group_cap_hint), 4096)
            except (HipSparkError, DeviceError) as e:
                self.dev.stop_recording()
                raise ExecutionError(str(e)) from e
            except BaseException:
                self.dev.stop_recording()
                raise
        raise ExecutionError("GROUP BY cardinality exceeds the on-chip aggregation tiers")

    def _switches(self) -> tuple:
        """The run-time switches a recording was made under (tests flip them between runs of one query)."""
        return (self.replay_enabled, self.short_tail_enabled, self.shared_tier_enabled, self.dict_enabled,
                self.fused_join_enabled, self.fused_probe_enabled, self.sharded_build_enabled, self.dev.zero_copy_results)

    @staticmethod
    def _stamp_of(path: str) -> tuple:
        try:
            st = os.stat(path)
            return (st.st_mtime_ns, st.st_size)
        except OSError:
            return (0, 0)

    def _file_stamps(self, plan: Any) -> list:
        """(path, stamp) of the table FILES the plan scans (attached device tables have no file behind them)."""
        out = []
        for key in getattr(plan, "_hs_scan_keys", None) or []:
            t = self._tables.get(key)
            if t is not None and t.stamp != ():
                out.append((key, self._stamp_of(key)))
        return out

    def _recording_key(self, plan: Any) -> Any:
        """A recorded run is valid for the same plan object over the same device buffers and capacities."""
        tables = []
        scans = getattr(plan, "_hs_scan_keys", None)
        if scans is None:
            scans = [str(Path(st.producer.file_path).resolve()) for st in plan.stages
                     if _cls(st.producer) == "LoadTableBlockTask"]
            plan._hs_scan_keys = scans
        for key in scans:
            t = self._tables.get(key)
            tables.append((_uid(t), tuple(sorted((cid, c.data.data_ptr()) for cid, c in t.columns.items()))) if t else None)
        return (_uid(plan), tuple(tables), self.group_cap_hint, self.merge_cap_hint, len(self._global_partial),
                len(self._global_merge))

    def _replay(self, rec: Any) -> list[JobResult] | None:
        self.dev.last_scan = getattr(rec, "scan_info", None)
        if not rec.self_cleaning:
            self.dev.flags.zero_()
        if not rec.replay():
            return None
        raw, nrows, flags = rec.finish()
        if flags:
            return None  # errors and dictionary growth are handled by the full path
        schema, stage_id = rec.result
        self.replays += 1
        return [self._emit_result(raw, nrows, schema, stage_id)]

    def _cached_plan(self, full_task: Any) -> Any:
        """Planning is pure in (task tree, table headers): keep the physical plan of the last few task
        trees.  Keyed by the root object (kept alive by the cache, so ids cannot be recycled) plus the
        stamps of the table files it scans."""
        stamps = []
        node, stack = full_task, []
        while node is not None and _cls(node) != "VoidTask":
            if _cls(node) == "LoadTableBlockTask":
                p = os.fspath(node.file_path)
                try:
                    st = os.stat(p)
                    stamps.append((p, node.alias, st.st_mtime_ns, st.st_size))
                except OSError:
                    stamps.append((p, node.alias, 0, 0))
            if _cls(node) == "BroadcastHashJoinTask":
                stack.append(node.right_side_task)
            node = node.parent_task
            if (node is None or _cls(node) == "VoidTask") and stack:
                node = stack.pop()
        key = (_uid(full_task), tuple(stamps))
        hit = self._plans.get(key)
        if hit is not None and hit[0] is full_task:
            plan = hit[1]
            for stage in plan.stages:
                stage.job_results.clear()
            return plan
        plan = self.generate_physical_plan(full_task)
        # the planner deep-copies the task tree: copies must not inherit the identity tokens of the originals
        # (a re-plan of the same DataFrame would otherwise reuse cache entries made for other file contents)
        for stage in plan.stages:
            for task in (stage.producer, *stage.consumers, stage.writer):
                getattr(task, "__dict__", {}).pop("_hs_uid", None)
        self._mark_short_tails(plan)
        if len(self._plans) >= 16:
            self._plans.pop(next(iter(self._plans)))
        self._plans[key] = (full_task, plan)
        return plan

    @staticmethod
    def _mark_short_tails(plan: Any) -> None:
        """Find [.. -> partial Aggregate -> shuffle write] feeding [shuffle read -> final Aggregate (-> Project)
        -> result write]: for these the work after the scan kernel can run as the two-launch short tail
        (Device.aggregate_partial(tail=True) + Device.aggregate_finish)."""
        for stage in plan.stages:
            consumers = list(stage.consumers)
            if _cls(stage.producer) != "LoadShuffleFilesTask" or _cls(stage.writer) != "WriteToLocalFileTask":
                continue
            if not consumers or _cls(consumers[0]) != "AggregateTask" or consumers[0].before_shuffle:
                continue
            if len(consumers) > 2 or (len(consumers) == 2 and _cls(consumers[1]) != "ProjectTask"):
                continue
            if len(stage.dependencies) != 1:
                continue
            dep = stage.dependencies[0]
            dep_consumers = list(dep.consumers)
            if (_cls(dep.writer) != "WriteToShufflePartitions" or not dep_consumers
                    or _cls(dep_consumers[-1]) != "AggregateTask" or not dep_consumers[-1].before_shuffle):
                continue
            dep_consumers[-1]._hs_short_tail = True

    # ---- stage execution -------------------------------------------------------------------------------
    def _run_stage(self, stage: Any, outputs: dict[int, Any]) -> list[JobResult]:
        from .device import SlabUnsupported, TierExceeded  # noqa: PLC0415

        producer, consumers, writer = stage.producer, list(stage.consumers), stage.writer
        kind = _cls(producer)
        if kind == "LoadTableBlockTask":
            ranges = self._stream_ranges(producer, consumers)
            if ranges is not None:
                if id(stage) in getattr(self, "_probe_stages", ()) and _cls(writer) == "WriteToShufflePartitions":
                    outputs[id(stage)] = _DeferredScan(stage, ranges)  # read by the join stage, range by range
                    self._job_seq += 1
                    return [JobResult(f"{self._job_prefix}-{self._job_seq}", self._executor_id, [])]
                return self._run_scan_stage_streamed(stage, outputs, ranges)
            batch = self._scan(producer, consumers, writer)
        elif kind == "LoadShuffleFilesTask":
            batch = outputs[id(stage.dependencies[0])]
            if batch.tail is not None:
                return [self._finish_short_tail(batch, consumers, writer, stage.stage_id)]
            if self.dist is not None:
                batch = self._exchange_partials(batch) if batch.slab is not None else self._exchange_partial_rows(batch)
        elif kind == "BroadcastHashJoinTask":
            first_real = next((t for t in consumers if _cls(t) != "FilterTask"), None)
            feeds_aggregate = first_real is not None and _cls(first_real) == "AggregateTask" and first_real.before_shuffle
            if isinstance(outputs[id(stage.dependencies[1])], _DeferredScan):
                return self._run_join_stage_streamed(stage, outputs, feeds_aggregate)
            batch = self._join(producer, outputs[id(stage.dependencies[0])], outputs[id(stage.dependencies[1])],
                               self._needed_names(consumers), feeds_aggregate, consumers)
        else:
            raise NotImplementedError(f"Job creation not implemented for {type(producer)}")

        batch = self._consume(batch, consumers)

        wname = _cls(writer)
        schema = writer.inferred_schema
        if wname == "WriteToShufflePartitions":
            # the "shuffle file" stays in HBM; data-dependent errors surface at the query's final read-back
            outputs[id(stage)] = batch if batch.tail is not None else self._quantise_batch(batch, schema)
            self._job_seq += 1
            return [JobResult(f"{self._job_prefix}-{self._job_seq}", self._executor_id, [])]
        if wname == "WriteToLocalFileTask":
            if self.dist is not None and batch.partitioned:
                batch = self._gather_to_root(batch)
            return [self._write_result(batch, schema, stage.stage_id)]
        raise NotImplementedError(f"writer {wname}")

    def _consume(self, batch: Any, consumers: Sequence[Any]) -> Any:
        """Run a stage's consumer tasks (filters, projections, aggregates) over one batch."""
        from .device import SlabUnsupported, TierExceeded  # noqa: PLC0415
        from .hipspark import HipSparkLimit  # noqa: PLC0415

        pending: list[Any] = []  # WHERE conditions not yet applied to `batch`
        for position, task in enumerate(consumers):
            tname = _cls(task)
            if tname == "FilterTask":
                pending.append(task.condition)
            elif tname == "ProjectTask":
                follower = consumers[position + 1] if position + 1 < len(consumers) else None
                if (pending and follower is not None and _cls(follower) == "AggregateTask" and follower.before_shuffle
                        and batch.unit_col is None):
                    lazy = self._project_keeping_filters(batch, pending, task)
                    if lazy is not None:  # the WHERE stays pending: the aggregate applies it while it scans
                        batch = lazy
                        continue
                batch = self._project(batch, pending, task)
                pending = []
            elif tname == "AggregateTask":
                if task.before_shuffle and batch.join8 is not None:
                    # the join's probe runs inside this aggregate's scan; what it cannot hold sends the
                    # query back through the materialising joins
                    try:
                        batch = self.dev.aggregate_join8(batch, pending, task.group_by_column, task.agg_columns,
                                                         task.inferred_schema, self.group_cap_hint,
                                                         cache_key=(_uid(task), "join8"), dist_ctx=self._dist_ctx())
                    except (TierExceeded, SlabUnsupported, HipSparkLimit):
                        self._no_join8.add(batch.join_task_id)
                        raise RestartQuery from None
                    batch.tail["task_id"] = _uid(task)
                    pending = []
                    continue
                if task.before_shuffle and self.dist is not None:
                    batch = self._agree_key_width(batch, task)  # the exchange form must not depend on local rows
                if task.before_shuffle and batch.unit_col is not None:
                    # rows of a join left in place, units = per-row partition ids: the shared-dictionary tier keys
                    # its tables on (unit, key); anything it cannot hold sends the query back through the general join
                    try:
                        cap = 16
                        while cap < max(self.group_cap_hint, 4) * batch.n_unit_ids:
                            cap *= 2
                        batch = self.dev.aggregate_partial(batch, pending, task.group_by_column, task.agg_columns,
                                                           task.inferred_schema, min(cap, SHARED_TIER_MAX),
                                                           cache_key=(_uid(task), "units"), shared=True)
                    except TierExceeded:
                        self._no_fused_join.add(batch.join_task_id)
                        raise RestartQuery from None
                    pending = []
                elif task.before_shuffle and (_uid(task) in self._global_partial or self.group_cap_hint > SHARED_TIER_MAX):
                    self._global_partial.add(_uid(task))
                    batch = self.dev.aggregate_partial_global(batch, pending, task.group_by_column, task.agg_columns,
                                                              task.inferred_schema)
                    batch.partitioned = self.dist is not None
                    pending = []
                elif task.before_shuffle and self.shared_tier_enabled and not self._private_tier_fits(task, batch, pending):
                    # tens to thousands of groups per unit: one LDS dictionary per workgroup, LDS atomics
                    try:
                        batch = self.dev.aggregate_partial(batch, pending, task.group_by_column, task.agg_columns,
                                                           task.inferred_schema, self.group_cap_hint,
                                                           cache_key=(_uid(task), "shared"), shared=True)
                    except TierExceeded:
                        self._global_partial.add(_uid(task))
                        batch = self.dev.aggregate_partial_global(batch, pending, task.group_by_column,
                                                                  task.agg_columns, task.inferred_schema)
                    batch.partitioned = self.dist is not None
                    pending = []
                elif task.before_shuffle:
                    slab_rows = None
                    if self.dist is not None:
                        from .distributed import max_local_units  # noqa: PLC0415

                        if batch.total_units is None:
                            raise NotImplementedError("multi-GPU aggregation needs a block-partitioned input")
                        slab_rows = max(batch.n_units, max_local_units(batch.total_units, self.world)) * self.group_cap_hint
                    if (self.short_tail_enabled and getattr(task, "_hs_short_tail", False)
                            and _uid(task) not in self._no_short_tail):
                        try:
                            batch = self.dev.aggregate_partial(batch, pending, task.group_by_column, task.agg_columns,
                                                               task.inferred_schema, self.group_cap_hint,
                                                               cache_key=(_uid(task), "tail"), slab_rows=slab_rows, tail=True)
                            batch.tail["task_id"] = _uid(task)
                            pending = []
                            continue
                        except (SlabUnsupported, TierExceeded):
                            self._no_short_tail.add(_uid(task))
                    try:
                        batch = self.dev.aggregate_partial(batch, pending, task.group_by_column, task.agg_columns,
                                                           task.inferred_schema, self.group_cap_hint,
                                                           cache_key=_uid(task), slab_rows=slab_rows)
                    except SlabUnsupported:
                        # e.g. variable-length string keys: partial rows go through the generic all-to-all
                        batch = self.dev.aggregate_partial(batch, pending, task.group_by_column, task.agg_columns,
                                                           task.inferred_schema, self.group_cap_hint,
                                                           cache_key=(_uid(task), "noslab"))
                    except TierExceeded:
                        try:  # private tables do not fit (many aggregates): the shared dictionary may
                            if not self.shared_tier_enabled:
                                raise
                            batch = self.dev.aggregate_partial(batch, pending, task.group_by_column, task.agg_columns,
                                                               task.inferred_schema, max(self.group_cap_hint, 16),
                                                               cache_key=(_uid(task), "shared"), shared=True)
                        except TierExceeded:
                            self._global_partial.add(_uid(task))
                            batch = self.dev.aggregate_partial_global(batch, pending, task.group_by_column,
                                                                      task.agg_columns, task.inferred_schema)
                    batch.partitioned = self.dist is not None and batch.slab is None
                    pending = []
                else:
                    batch = self._materialise(batch, pending)
                    pending = []
                    was_partitioned = batch.partitioned
                    use_global = (_uid(task) in self._global_merge or self._merge_overflowed
                                  or (batch.partitioned and batch.order is not None))
                    if not use_global:
                        try:
                            batch = self.dev.aggregate_merge(batch, task.agg_columns, task.inferred_schema,
                                                             self.merge_cap_hint)
                            self._lds_merges.append(_uid(task))  # HS_FLAG_MERGE_ROWS sends them to the HBM tier
                        except TierExceeded:
                            use_global = True
                    if use_global:
                        if not was_partitioned:
                            self._global_merge.add(_uid(task))
                        batch = self.dev.aggregate_merge_global(batch, task.agg_columns, task.inferred_schema)
                    batch.partitioned = was_partitioned
            else:
                raise NotImplementedError(f"consumer {tname}")
        return self._materialise(batch, pending)

    # ---- producers -------------------------------------------------------------------------------------
    # ---- tables beyond HBM: the scan stage in block ranges -------------------------------------------------
    def _scan_columns(self, table: Any, alias: str, consumers: Sequence[Any]) -> list[int]:
        prefix = f"{alias}." if alias else ""
        names = [prefix + n for n, _ in table.schema]
        needed = self._needed_names(consumers)
        col_ids = list(range(len(names))) if needed is None else sorted({names.index(n) for n in needed if n in names})
        return col_ids or [0]

    def _stream_ranges(self, producer: Any, consumers: Sequence[Any]) -> list[list[int]] | None:
        """Block ranges to stream a scan in, or None when the referenced columns fit the budget (the usual case: they
        are loaded once and stay resident).  Budget = engine.hbm_budget bytes (HIPSPARK_HBM_BUDGET), default half of
        the free device memory; a range holds as many consecutive blocks as fit half of it."""
        from . import table as tbl  # noqa: PLC0415

        if self.dist is not None:
            # Streaming is a per-rank decision (local free memory, local blocks) while the resident path's width
            # agreements and the result gather are collectives: ranks deciding differently would issue mismatched
            # collectives, and a streamed to-file stage would keep every rank's rows in its own file.  On N ranks a
            # rank holds 1/N of the table and stays resident.
            return None
        key = str(Path(producer.file_path).resolve())
        cached = self._tables.get(key)
        if cached is not None and cached.stamp == ():
            return None  # attached device tables are resident by definition
        table = self._table(producer.file_path)
        col_ids = self._scan_columns(table, producer.alias, consumers)
        if all(c in table.columns for c in col_ids):
            return None
        budget = self.hbm_budget
        if budget is None:
            import torch  # noqa: PLC0415

            budget = torch.cuda.mem_get_info(self.dev.device)[0] // 2
        per_block = tbl.referenced_block_bytes(table, col_ids)
        if sum(per_block) <= budget:
            return None
        ranges, cur, cur_bytes = [], [], 0
        for b, nbytes in enumerate(per_block):
            if cur and cur_bytes + nbytes > budget // 2:
                ranges.append(cur)
                cur, cur_bytes = [], 0
            cur.append(b)
            cur_bytes += nbytes
        if cur:
            ranges.append(cur)
        return ranges

    def _run_scan_stage_streamed(self, stage: Any, outputs: dict[int, Any], ranges: list[list[int]]) -> list[JobResult]:
        """A scan stage over a table that does not fit HBM: block range by block range through the ingest pipeline
        (pruned reads -> pinned staging -> HBM), the stage's consumers run on every range, and
        * partial-aggregate rows (the stage's "shuffle file") accumulate on the device across the ranges - units stay
          file blocks, so the final merge sees exactly the rows and the order of the resident run;
        * rows of a non-aggregating stage are appended, range after range, to the result BlockFile with the reference's
          append-merge rule (io.py:231-252: a short last block is re-written, then new blocks follow), so the file
          stays readable by the reference.
        The range's columns are freed before the next range is read."""
        from . import table as tbl  # noqa: PLC0415
        from .device import DBatch  # noqa: PLC0415

        producer, consumers, writer = stage.producer, list(stage.consumers), stage.writer
        table = self._table(producer.file_path)
        col_ids = self._scan_columns(table, producer.alias, consumers)
        if self.dev.rec is not None:
            self.dev.rec.poisoned = True  # host data flows in on every run
        for task in consumers:  # the one-launch tail wants every unit's slab rows in ONE launch
            if _cls(task) == "AggregateTask" and task.before_shuffle:
                self._no_short_tail.add(_uid(task))
        schema = writer.inferred_schema
        to_file = _cls(writer) == "WriteToLocalFileTask"
        parts: list[Any] = []
        out_path, rows_written = None, 0
        self.streamed_ranges += len(ranges)
        for blocks in ranges:
            sub = tbl.sub_table(table, blocks)
            tbl.load_columns(self.dev, sub, col_ids)
            batch = tbl.table_batch(sub, col_ids, producer.alias)
            batch.partitioned = self.dist is not None
            batch = self.dev.resolve(self._consume(batch, consumers))
            quantised = self.dev.resolve(self._quantise_batch(batch, schema))
            if not to_file:
                parts.append(quantised)
                continue
            raw, nrows, flags = self.dev.download_batch(quantised, schema, None)
            self._act_on_flags(flags)  # also HS_FLAG_DICT_FULL: the query is repeated with larger dictionaries
            if nrows:
                if out_path is None:
                    out_path = self._result_path(stage.stage_id)
                    out_path.parent.mkdir(parents=True, exist_ok=True)
                    out_path.unlink(missing_ok=True)
                BlockFile(out_path, list(schema)).append_raw(raw)
                rows_written += nrows
            del sub, batch, quantised
        self._job_seq += 1
        job_id = f"{self._job_prefix}-{self._job_seq}"
        if to_file:
            from .jobs import OutputFile  # noqa: PLC0415

            files = [OutputFile(out_path)] if rows_written and self.rank == 0 else []
            return [JobResult(job_id, self._executor_id, files)]
        n = sum(p.nrows for p in parts)
        cols = [self.dev.concat_cols([p.cols[c] for p in parts]) for c in range(len(schema))] if parts else []
        order = None
        if parts and all(p.order is not None for p in parts):
            import torch  # noqa: PLC0415

            order = torch.cat([p.order[: p.nrows] for p in parts])
        merged = DBatch(list(schema), cols, n, [0, n], order=order,
                        total_units=table.total_blocks if table.total_blocks is not None else len(table.block_rows),
                        partitioned=self.dist is not None)
        outputs[id(stage)] = merged
        return [JobResult(job_id, self._executor_id, [])]

    def _run_join_stage_streamed(self, stage: Any, outputs: dict[int, Any], feeds_aggregate: bool) -> list[JobResult]:
        """A join whose probe (right) side does not fit HBM (reference: the right side is streamed block by
        block through the build side's hash map, tasks.py:224-240).  The build side is resident; the probe side's scan
        stage was deferred and runs here range by range: read (pruned, pipelined) -> its own consumers -> join ->
        * join feeding a GROUP BY (the byte-table join with the probe inside the aggregate): the table is
          built ONCE; every range leaves the RAW per-JoinJob tables; they are added up in range order before the one
          rounding the reference applies per JoinJob, then the short tail runs as for a resident table;
        * join feeding the result file: the joined rows of every range go through the stage's consumers and are appended
          to the result BlockFile with the reference's append-merge rule (io.py:231-252).
        One rank only (a streamed stage is a per-rank decision, see _stream_ranges)."""
        from . import table as tbl  # noqa: PLC0415
        from .device import SlabUnsupported, TierExceeded  # noqa: PLC0415
        from .hipspark import HipSparkLimit  # noqa: PLC0415

        producer, consumers, writer = stage.producer, list(stage.consumers), stage.writer
        left = outputs[id(stage.dependencies[0])]
        deferred = outputs[id(stage.dependencies[1])]
        scan = deferred.stage
        table = self._table(scan.producer.file_path)
        scan_consumers = list(scan.consumers)
        col_ids = self._scan_columns(table, scan.producer.alias, scan_consumers)
        needed = self._needed_names(consumers)
        if self.dev.rec is not None:
            self.dev.rec.poisoned = True  # host data flows in on every run
        to_file = _cls(writer) == "WriteToLocalFileTask"
        if not feeds_aggregate and not to_file:
            raise ExecutionError("a join over a probe side larger than the HBM budget must feed a GROUP BY or the result file")
        self.streamed_ranges += len(deferred.ranges)
        agg = next((t for t in consumers if _cls(t) == "AggregateTask"), None)
        raw_tables: list = []
        prepared = None
        out_path, rows_written = None, 0
        schema = writer.inferred_schema
        self._join8_reuse = {}
        left = self.dev.resolve(left)
        for blocks in deferred.ranges:
            sub = tbl.sub_table(table, blocks)
            tbl.load_columns(self.dev, sub, col_ids)
            if self.dict_enabled:
                self._encode_string_columns(sub, col_ids)
            right = tbl.table_batch(sub, col_ids, scan.producer.alias)
            right = self.dev.resolve(self._quantise_batch(self.dev.resolve(self._consume(right, scan_consumers)), scan.writer.inferred_schema))
            joined = self._join(producer, left, right, needed, feeds_aggregate, consumers)
            if feeds_aggregate:
                if joined.join8 is None:
                    raise ExecutionError("this join shape cannot stream its probe side: only the byte-table join (unique dense "
                                         "INTEGER build keys, at most one dictionary-coded build column) adds partial "
                                         "aggregates up across block ranges before the per-JoinJob rounding")
                pending = [t.condition for t in consumers[: consumers.index(agg)] if _cls(t) == "FilterTask"]
                if any(_cls(t) not in ("FilterTask",) for t in consumers[: consumers.index(agg)]):
                    raise ExecutionError("a projection between a streamed join and its GROUP BY is not supported")
                try:
                    p = self.dev.aggregate_join8(joined, pending, agg.group_by_column, agg.agg_columns, agg.inferred_schema,
                                                 self.group_cap_hint, cache_key=None, raw_tables=raw_tables)
                except (TierExceeded, SlabUnsupported, HipSparkLimit) as e:
                    raise ExecutionError(f"streamed join: {e}") from e
                if prepared is not None and (p["unit_cap"], p["slots"], p["table_bytes"]) != (
                        prepared["unit_cap"], prepared["slots"], prepared["table_bytes"]):
                    raise ExecutionError("streamed join: the ranges' unit tables differ in shape")
                prepared = p
            else:
                batch = self.dev.resolve(self._consume(joined, consumers))
                raw, nrows, flags = self.dev.download_batch(self.dev.resolve(self._quantise_batch(batch, schema)), schema, None)
                self._act_on_flags(flags)
                if nrows:
                    if out_path is None:
                        out_path = self._result_path(stage.stage_id)
                        out_path.parent.mkdir(parents=True, exist_ok=True)
                        out_path.unlink(missing_ok=True)
                    BlockFile(out_path, list(schema)).append_raw(raw)
                    rows_written += nrows
            del sub, right, joined
        self._join8_reuse = None
        self._job_seq += 1
        job_id = f"{self._job_prefix}-{self._job_seq}"
        if not feeds_aggregate:
            from .jobs import OutputFile  # noqa: PLC0415

            return [JobResult(job_id, self._executor_id, [OutputFile(out_path)] if rows_written else [])]
        batch = self.dev.join8_finish_ranges(prepared, raw_tables, agg.inferred_schema)
        batch.tail["task_id"] = _uid(agg)
        outputs[id(stage)] = batch
        return [JobResult(job_id, self._executor_id, [])]

    def _result_path(self, stage_id: str) -> Path:
        if self._result_root is None:
            base = Path(self._work_folder) if self._work_folder else constants.SHUFFLE_FOLDER
            self._result_root = base / f"hip-{uuid.uuid4().hex[:12]}"
            self._owned_dirs.add(self._result_root)
        out_file = self._result_paths.get(stage_id)
        if out_file is None:
            out_file = self._result_paths[stage_id] = self._result_root / str(stage_id) / "result.bin"
        return out_file

    def _scan(self, producer: Any, consumers: Sequence[Any], writer: Any) -> Any:
        from . import table as tbl  # noqa: PLC0415

        table = self._table(producer.file_path)
        prefix = f"{producer.alias}." if producer.alias else ""
        names = [prefix + n for n, _ in table.schema]
        needed: list[str] | None = []
        for task in consumers:
            tname = _cls(task)
            if tname == "FilterTask":
                needed += _plain_names(task.condition)
            elif tname == "ProjectTask":
                for col in task.columns:
                    needed += _plain_names(col)
                break
            elif tname == "AggregateTask":
                needed += _plain_names(task.group_by_column)
                for agg in task.agg_columns:
                    needed += _plain_names(agg)
                break
        else:
            needed = None  # rows reach the writer unprojected: every column is needed
        col_ids = list(range(len(names))) if needed is None else sorted({names.index(n) for n in needed})
        if not col_ids:
            col_ids = [0]
        tbl.load_columns(self.dev, table, col_ids)
        if self.dist is not None:
            self._agree_table_widths(table, col_ids)
            if self.dict_enabled:
                self._encode_string_columns_on_ranks(table, col_ids)
        elif self.dict_enabled:
            self._encode_string_columns(table, col_ids)
        batch = tbl.table_batch(table, col_ids, producer.alias)
        batch.partitioned = self.dist is not None
        return batch

    def _encode_string_columns(self, table: Any, col_ids: Sequence[int]) -> None:
        """Table open, string columns with few distinct values: one code byte per row + the dictionary (DCol.dict).
        Tried once per column; the plain column stays attached (DCol.plain) for whoever needs the bytes.  (Single-GPU
        engines only: ranks would have to agree on one dictionary per column first.)"""
        from . import hipspark as hs  # noqa: PLC0415

        tried = table.__dict__.setdefault("_hs_dict_tried", set())
        for cid in col_ids:
            col = table.columns[cid]
            if cid in tried or col.kind != hs.STR:
                continue
            tried.add(cid)
            coded = self.dev.dict_encode(col)
            if coded is not None:
                table.columns[cid] = coded

    def _encode_string_columns_on_ranks(self, table: Any, col_ids: Sequence[int]) -> None:
        """_encode_string_columns when every rank holds other blocks of the table: the ranks code their own rows, then
        AGREE on one dictionary per column - all-gather of the (at most 256) distinct strings, union, sorted - and
        re-code their bytes through a 256-entry table.  Equal codes then mean equal strings on every rank: code bytes
        may travel (the join's gathered build side, exchange slabs) and per-dictionary predicate bits are the same
        everywhere.  A column that any rank cannot code, or whose union outgrows a code byte, stays plain on ALL ranks
        (once per table and column; every rank scans the same tables with the same column lists, so the collectives
        match)."""
        from . import hipspark as hs  # noqa: PLC0415

        tried = table.__dict__.setdefault("_hs_dict_tried", set())
        for cid in sorted(col_ids):
            col = table.columns[cid]
            if cid in tried or col.kind != hs.STR:
                continue
            tried.add(cid)
            coded = self.dev.dict_encode(col) if col.n > 0 else None
            mine = [] if col.n == 0 else (list(coded.dict) if coded is not None else None)
            everyone: list = [None] * self.world
            self.dist.all_gather_object(everyone, mine, group=self.group)
            if any(entries is None for entries in everyone):
                continue
            union = sorted(set().union(*map(set, everyone)))
            if not union or len(union) > 256:
                continue
            table.columns[cid] = self.dev.dict_recode(col, coded, tuple(union))

    def _agree_table_widths(self, table: Any, col_ids: Sequence[int]) -> None:
        """fixed_len of a stored STRING column becomes a FILE-global property (once per table and column): see
        _agree_key_width.  Every rank scans the same tables with the same column lists, so the collectives match."""
        from . import hipspark as hs  # noqa: PLC0415
        from .distributed import agree_string_width  # noqa: PLC0415

        agreed = table.__dict__.setdefault("_hs_widths_agreed", set())
        for cid in col_ids:
            col = table.columns[cid]
            if col.kind != hs.STR or cid in agreed:
                continue
            width = agree_string_width(self.dist, col.fixed_len, col.n, self.dev.device, self.group)
            table.columns[cid] = self.dev.with_string_width(col, width)
            table.columns[cid]._hs_width_agreed = True  # (see _agree_key_width)
            agreed.add(cid)

    @staticmethod
    def _needed_names(consumers: Sequence[Any]) -> set[str] | None:
        """Column names the consumers of a producer reference before the schema changes (projection /
        aggregate); None = rows reach the writer as they are, every column is needed."""
        needed: set[str] = set()
        for task in consumers:
            tname = _cls(task)
            if tname == "FilterTask":
                needed.update(_plain_names(task.condition))
            elif tname == "ProjectTask":
                for col in task.columns:
                    needed.update(_plain_names(col))
                return needed
            elif tname == "AggregateTask":
                needed.update(_plain_names(task.group_by_column))
                for agg in task.agg_columns:
                    needed.update(_plain_names(agg))
                return needed
        return None

    def _dist_ctx(self) -> tuple | None:
        return (self.dist, self.group, self.world) if self.dist is not None else None

    def _join_fused_probe(self, task: Any, left: Any, right: Any, lkey: int, rkey: int, needed: set[str],
                          consumers: Sequence[Any]) -> Any:
        """Primary-key / foreign-key join feeding a partial aggregate, round 3 (BASELINE config 4 on 1..N GPUs; DESIGN.md
        4.6): the build side becomes a BYTE table (slot = key - key_min, value = the dictionary code of the one
        build-side column the aggregate reads) and the probe happens inside the aggregate's scan - per probe row nothing
        is written at all.  On N ranks the (small) build side is all-gathered and every rank builds the whole table;
        probe rows never cross xGMI.  -> a batch of virtual columns for Device.aggregate_join8, or None when the shape
        does not qualify (decided from plan structure and GLOBAL table properties only, so every rank decides alike)."""
        from . import hipspark as hs  # noqa: PLC0415
        from .device import DBatch, DCol  # noqa: PLC0415

        agg = next((t for t in consumers if _cls(t) != "FilterTask"), None)
        if (agg is None or not getattr(agg, "_hs_short_tail", False) or _uid(agg) in self._no_short_tail
                or not self.short_tail_enabled or not self.shared_tier_enabled or left.lazy or right.lazy):
            return None
        if left.cols[lkey].kind != hs.I32 or right.cols[rkey].kind != hs.I32 or constants.SHUFFLE_PARTITIONS > 127:
            return None
        left_names = {name for name, _ in left.schema}
        for t in consumers:  # a WHERE between join and aggregate may look at the probe side only
            if _cls(t) == "FilterTask" and any(name in left_names for name in _plain_names(t.condition)):
                return None
        wanted_left = [i for i, (name, _) in enumerate(left.schema) if name in needed]
        if len(wanted_left) > 1:
            return None
        payload = left.cols[wanted_left[0]] if wanted_left else None
        if payload is not None and (payload.dict is None or len(payload.dict) > 255):
            return None
        dev = self.dev
        shape = dev.join8_plan(left.cols[lkey], self._dist_ctx())
        if shape is None or right.nrows >= 0xFFFFFFFF:
            return None
        if right.cols[rkey].data.data_ptr() % 16 or left.cols[lkey].data.data_ptr() % 16:
            raise ExecutionError("join key columns must be 16-byte aligned")
        ctx = self._dist_ctx()
        stripes = None
        if ctx is not None and self.sharded_build_enabled and ctx[2] <= 64:
            # N ranks: a probe table clustered on the key (every block a key stripe) lets every rank build only the part of
            # the table its own blocks can reach, from build rows routed to it; otherwise the whole build side is gathered
            stripes = dev.join8_stripes(right.cols[rkey], right.unit_rows, right.unit_ids, shape, ctx)
        reuse = getattr(self, "_join8_reuse", None)  # a streamed probe side: the table is built for the first range only
        if reuse is not None and reuse.get("for") is left.cols[lkey]:
            j = dict(reuse["table"])
        elif stripes is not None:
            j = dev.join8_table_sharded(shape, stripes, left.cols[lkey], payload, constants.SHUFFLE_PARTITIONS, ctx)
            self.sharded_builds += 1
        else:
            j = dev.join8_table(shape, left.cols[lkey], payload, constants.SHUFFLE_PARTITIONS, ctx)
        if reuse is not None and reuse.get("for") is None:
            reuse["for"], reuse["table"] = left.cols[lkey], dict(j)
        j["probe_key"] = right.cols[rkey]
        n = right.nrows
        schema, cols = [], []
        if payload is not None:
            schema.append(left.schema[wanted_left[0]])
            cols.append(DCol(hs.STR, right.cols[rkey].data, n, lens=dev.const_lens(1, n), offs=None, fixed_len=1,
                             dict=payload.dict, virtual=hs.JOIN8_CODE))
        for (name, ctype), col in zip(right.schema, right.cols):
            if name in needed:
                schema.append((name, ctype))
                cols.append(col)
        if len(cols) == (1 if payload is not None else 0):  # e.g. COUNT only: a row count carrier
            schema.append(right.schema[rkey])
            cols.append(right.cols[rkey])
        joined = DBatch(schema, cols, n, [0, n])
        joined.join8 = j
        joined.join_task_id = _uid(task)
        joined.total_units = constants.SHUFFLE_PARTITIONS
        self._fused_join_tasks.add(_uid(task))
        self.fused_joins += 1
        self.fused_probes += 1
        return joined

    def _join(self, task: Any, left: Any, right: Any, needed: set[str] | None = None,
              feeds_aggregate: bool = False, consumers: Sequence[Any] = ()) -> Any:
        """Partitioned inner hash join; output rows grouped by ``hash(key) % SHUFFLE_PARTITIONS`` so
        that a following partial aggregate sees the reference's JoinJob units (plan.py:99-109)."""
        from . import hipspark as hs  # noqa: PLC0415
        from .device import DBatch  # noqa: PLC0415

        left, right = self.dev.resolve(left), self.dev.resolve(right)
        lkey = left.column_index(task.left_key.name)
        rkey = right.column_index(task.right_key.name)
        # codes of two dictionaries do not compare: join keys are matched on the strings themselves
        left, right = self.dev.decoded_batch(left, [lkey]), self.dev.decoded_batch(right, [rkey])
        if (feeds_aggregate and self.fused_join_enabled and self.fused_probe_enabled and needed is not None
                and _uid(task) not in self._no_fused_join and _uid(task) not in self._no_join8):
            fused = self._join_fused_probe(task, left, right, lkey, rkey, needed, consumers)
            if fused is not None:
                return fused
        if (feeds_aggregate and self.fused_join_enabled and self.dist is None and needed is not None
                and _uid(task) not in self._no_fused_join and left.nrows > 0 and right.nrows > 0
                and left.cols[lkey].kind == hs.I32 and right.cols[rkey].kind == hs.I32
                and left.nrows < 0xFFFFFFFF and constants.SHUFFLE_PARTITIONS <= 127
                and right.cols[rkey].data.data_ptr() % 16 == 0 and left.cols[lkey].data.data_ptr() % 16 == 0):
            return self._join_in_place(task, left, right, lkey, rkey, needed)
        if self.dist is not None:
            # both inputs travel to the owner of their key's partition (p % world), then a local join
            left, _ = self._exchange_by_key(left, lkey)
            right, rpart = self._exchange_by_key(right, rkey)
            perm, part_start = self.dev.partition_by_ids(rpart, right.nrows, constants.SHUFFLE_PARTITIONS)
        else:
            perm, part_start = self.dev.partition(right, rkey, constants.SHUFFLE_PARTITIONS)
        right = self.dev.gather_batch(right, perm, right.nrows, part_start)
        out_left, out_right, out_start, n_out = self.dev.join_indices(left.cols[lkey], right.cols[rkey])
        if self.dev.rec is not None:
            self.dev.rec.poisoned = True  # unit boundaries of the joined rows are learnt on the host below
        starts = out_start.tolist() if right.nrows <= 4096 else None
        if starts is None:
            import torch  # noqa: PLC0415

            idx = torch.tensor(part_start, dtype=torch.int64, device=out_start.device)
            unit_rows = [int(v) for v in out_start[idx].tolist()]
        else:
            unit_rows = [int(starts[p]) for p in part_start]
        # column pruning: the reference materialises every column of both sides for every matching pair
        # (tasks.py:229-238); only the columns a downstream operator names are gathered here
        schema, cols = [], []
        for side, idx in ((left, out_left), (right, out_right)):
            for (name, ctype), col in zip(side.schema, side.cols):
                if needed is None or name in needed:
                    schema.append((name, ctype))
                    cols.append(self.dev.gather_col(col, idx, n_out))
        if not cols:  # e.g. COUNT only: keep one column so the batch has a row count carrier
            schema.append(right.schema[rkey])
            cols.append(self.dev.gather_col(right.cols[rkey], out_right, n_out))
        joined = DBatch(schema, cols, n_out, unit_rows)
        if self.dist is not None:  # units = shuffle partitions, the same ids on every rank
            joined.unit_ids = list(range(constants.SHUFFLE_PARTITIONS))
            joined.total_units = constants.SHUFFLE_PARTITIONS
            joined.partitioned = True
        return joined

    def _agree_key_width(self, batch: Any, task: Any) -> Any:
        """Multi-GPU: which exchange a partial aggregate uses (slab all-gather / short tail vs the generic
        all-to-all) depends on whether its STRING key has a fixed width.  ``fixed_len`` of a column comes from the
        LOCAL rows (0 on a rank that owns no block), so the ranks agree on the global width first and describe
        their key column with it: every rank then raises SlabUnsupported - or none does."""
        import dataclasses  # noqa: PLC0415

        from . import hipspark as hs  # noqa: PLC0415
        from .distributed import agree_string_width  # noqa: PLC0415
        from .lowering import unalias  # noqa: PLC0415

        try:
            idx = batch.column_index(unalias(task.group_by_column).name)
        except (ValueError, AttributeError):
            return batch  # reported by the lowering with the reference's error
        col = batch.cols[idx]
        if col.kind != hs.STR:
            return batch
        if col.dict is not None or getattr(col, "_hs_width_agreed", False):
            # one code byte per row under a dictionary all ranks share, or a table column whose width was agreed when the
            # table was opened: the width IS global - no collective per run.  (Round 3: this collective is not part of a
            # recorded run; a rank replaying its recording while another took the full path - recordings become available
            # at rank-local moments - left the other rank alone in it: fuzz seed 610 on 2 ranks hung.)
            return batch
        if batch.lazy:
            batch = self.dev.resolve(batch)
            col = batch.cols[idx]
        if self.dev.rec is not None:
            self.dev.rec.poisoned = True  # a run with a collective of its own is never replayed, on any rank
        width = agree_string_width(self.dist, col.fixed_len, col.n, self.dev.device, self.group)
        if width == col.fixed_len:
            return batch
        cols = list(batch.cols)
        cols[idx] = self.dev.with_string_width(col, width)
        return dataclasses.replace(batch, cols=cols)

    def _join_in_place(self, task: Any, left: Any, right: Any, lkey: int, rkey: int, needed: set[str]) -> Any:
        """Primary-key / foreign-key join feeding a partial aggregate (BASELINE config 4): the probe
        side's rows stay where they are - no partitioning pass, no pair lists, no gather of its columns.  Per probe
        row the probe kernel emits the matching build row and the row's shuffle partition hash(key) % 10 as a UNIT
        id; the aggregate that follows keys its tables on (unit, group key), which yields exactly the reference's
        per-JoinJob partial rows (plan.py:99-109).  Build-side columns the aggregate names are gathered by the build
        row (a dictionary-coded one rides along inside the probe kernel, one byte per row).  Duplicate build keys
        raise HS_FLAG_JOIN_DUP: the query is then re-run through the general join."""
        from .device import DBatch, DCol  # noqa: PLC0415

        dev = self.dev
        wanted_left = [i for i, (name, _) in enumerate(left.schema) if name in needed]
        payload_idx = next((i for i in wanted_left if left.cols[i].dict is not None), None)
        others = [i for i in wanted_left if i != payload_idx]
        rows, unit, pay = dev.join_probe_unique(left.cols[lkey], right.cols[rkey], constants.SHUFFLE_PARTITIONS,
                                                payload=left.cols[payload_idx] if payload_idx is not None else None,
                                                want_rows=bool(others))
        schema, cols = [], []
        n = right.nrows
        for i in wanted_left:
            schema.append(left.schema[i])
            if i == payload_idx:
                src = left.cols[i]
                cols.append(DCol(src.kind, pay, n, lens=dev.const_lens(1, n), offs=None, fixed_len=1, dict=src.dict))
            else:
                cols.append(dev.gather_col(left.cols[i], rows, n))  # rows without a match read build row 0: dropped
        for (name, ctype), col in zip(right.schema, right.cols):
            if name in needed:
                schema.append((name, ctype))
                cols.append(col)
        if not cols:  # e.g. COUNT only: a row count carrier
            schema.append(right.schema[rkey])
            cols.append(right.cols[rkey])
        joined = DBatch(schema, cols, n, [0, n])
        joined.unit_col, joined.n_unit_ids = unit, constants.SHUFFLE_PARTITIONS
        joined.join_task_id = _uid(task)
        self._fused_join_tasks.add(_uid(task))
        self.fused_joins += 1
        return joined

    def _exchange_partials(self, batch: Any) -> Any:
        """The shuffle between the two aggregation phases on N GPUs: all-gather the fixed-size slabs, then
        un-interleave the columns.  Buffers are allocated here once per run and the collective + unpack is one
        recorded operation, so a replayed query repeats exactly this exchange."""
        import torch  # noqa: PLC0415

        from . import hipspark as hs  # noqa: PLC0415
        from .device import DBatch, DCol  # noqa: PLC0415
        from .distributed import all_gather_slabs_into  # noqa: PLC0415

        if batch.slab is None:
            raise NotImplementedError("multi-GPU exchange of this stage's output is not built yet")
        layout = batch.slab_layout
        world, m = self.world, layout.slab_rows
        n = world * m
        device = batch.slab.device
        gathered = torch.empty(world * layout.nbytes, dtype=torch.uint8, device=device)
        flags = torch.empty(world, dtype=torch.int32, device=device)
        order = torch.empty(n, dtype=torch.int64, device=device)
        cols = [torch.empty(n * c.row_bytes // torch.empty((), dtype=c.dtype).element_size() + 16, dtype=c.dtype,
                            device=device)[: n * c.row_bytes // torch.empty((), dtype=c.dtype).element_size()]
                for c in layout.columns]
        slab, dist, group = batch.slab, self.dist, self.group
        # one collective (recorded as an opaque step) + ONE library launch that un-interleaves the slabs
        self.dev.op(all_gather_slabs_into, dist, slab, gathered, group)
        import ctypes as C  # noqa: PLC0415

        ncols = len(layout.columns)
        offs = (C.c_int64 * ncols)(*[c.offset for c in layout.columns])
        rbytes = (C.c_int32 * ncols)(*[c.row_bytes for c in layout.columns])
        dsts = (C.c_void_p * ncols)(*[t.data_ptr() for t in cols])
        hs.check(self.dev.lib.hs_slab_unpack(self.dev.stream, gathered.data_ptr(), world, layout.nbytes, m,
                                             layout.order_offset, ncols, offs, rbytes, dsts, flags.data_ptr(),
                                             order.data_ptr()), "hs_slab_unpack")
        if self.dev.rec is not None:
            self.dev.rec.keep.append((gathered, flags, order, cols))
        self._remote_flags = flags
        out = []
        for src, slab_col in zip(batch.cols, batch.slab_cols):
            data = cols[slab_col]
            if src.kind == hs.STR:
                out.append(DCol(hs.STR, data, n, lens=self.dev.const_lens(src.fixed_len, n), offs=None,
                                fixed_len=src.fixed_len, dict=src.dict))
            else:
                out.append(DCol(src.kind, data, n))
        return DBatch(list(batch.schema), out, n, [0, n], None, order=order, total_units=batch.total_units)

    def _exchange_rows(self, batch: Any, dest: Any, extras: Sequence[Any] = ()) -> tuple[Any, list[Any]]:
        """Generic shuffle (reference tasks.py:347-375 routes every row by hash(key) % 10 through files): row i goes to
        rank dest[i]; `extras` are per-row tensors (u8 / i64) that travel along.  TWO collectives whatever the number
        of columns: one small all-to-all of a size matrix (rows and string-payload bytes per peer - the split sizes
        of the data collective must be host integers), then ONE all_to_all_single of a byte buffer in which every
        destination's share holds its slice of every column back to back (RCCL: direct peer-to-peer transfers on the
        xGMI mesh).  Round 3: the byte buffer is packed and unpacked ON THE DEVICE - stable counting sort by
        destination, one gather per column, then ONE launch that copies the world x pieces slices into place
        (hs_copy_segments; round 2 sliced and concatenated torch tensors in Python loops) - and the sizes the host needs
        (rows and payload bytes per destination) come back in a single read.  Returns (batch of received rows grouped
        by source rank, received extras)."""
        from . import hipspark as hs  # noqa: PLC0415
        from .device import DBatch, DCol  # noqa: PLC0415
        from .distributed import all_to_all_rows, exchange_size_matrix  # noqa: PLC0415

        dev, world, dist, group = self.dev, self.world, self.dist, self.group
        batch = dev.decoded_batch(dev.resolve(batch))  # codes of per-rank dictionaries do not travel
        n = batch.nrows
        self._generic_exchange_used = True
        perm, start_dev = dev.partition_by_ids_dev(dest, n, world)
        # every column (and extra) permuted into destination order; pieces = (bytes tensor, bytes per row or None for a
        # variable-length payload whose per-destination sizes come from its offsets)
        pieces: list[tuple[Any, int | None, Any]] = []
        layout: list[tuple[str, Any]] = []  # how to rebuild the columns on the receiving side
        for col in batch.cols:
            g = dev.permute_col(col, perm, n)
            if g.kind == hs.STR:
                pieces.append((g.lens, 1, None))
                pieces.append((g.data, g.fixed_len if g.fixed_len >= 0 else None, g.offs))
                layout.append(("str", None))
            else:
                pieces.append((g.data, hs.KIND_BYTES[g.kind], None))
                layout.append(("fixed", g.kind))
        for t in extras:
            kind = hs.U8 if t.dtype == dev.torch.uint8 else hs.I64
            g = dev.permute_col(DCol(kind, t, n), perm, n)
            pieces.append((g.data, hs.KIND_BYTES[kind], None))
            layout.append(("extra", kind))
        # ONE read-back: destination boundaries + the payload offsets of every variable-length piece at them
        start, var_offs = dev.exchange_boundaries(start_dev, world, [p[2] for p in pieces if p[1] is None])
        rows_to = [start[d + 1] - start[d] for d in range(world)]
        piece_bytes: list[list[int]] = []  # [piece][destination]
        piece_base: list[list[int]] = []   # byte offset of the destination's slice inside the piece
        v = 0
        for _data, width, _offs in pieces:
            if width is None:
                bounds = var_offs[v]
                v += 1
                piece_base.append([int(bounds[d]) for d in range(world)])
                piece_bytes.append([int(bounds[d + 1] - bounds[d]) for d in range(world)])
            else:
                piece_base.append([start[d] * width for d in range(world)])
                piece_bytes.append([rows_to[d] * width for d in range(world)])
        # sizes: per peer [rows, payload bytes of every string column] (a column looks fixed-width from local rows only)
        str_payload = [i for i, (tag, _) in enumerate(self._piece_tags(layout)) if tag == "payload"]
        mine = [[rows_to[d]] + [piece_bytes[i][d] for i in str_payload] for d in range(world)]
        theirs = exchange_size_matrix(dist, mine, dev.device, group)
        rows_from = [row[0] for row in theirs]
        n_in = sum(rows_from)
        recv_piece_bytes: list[list[int]] = []  # [piece][source]
        k = 0
        for tag, arg in self._piece_tags(layout):
            if tag == "payload":
                recv_piece_bytes.append([row[1 + k] for row in theirs])
                k += 1
            else:
                recv_piece_bytes.append([c * arg for c in rows_from])
        send_splits = [sum(pb[d] for pb in piece_bytes) for d in range(world)]
        recv_splits = [sum(pb[s] for pb in recv_piece_bytes) for s in range(world)]
        # pack: destination-major, piece-minor - one launch
        send_buf = dev.empty(sum(send_splits), dev.torch.uint8)
        segs, pos = [], 0
        for d in range(world):
            for i, (data, _w, _o) in enumerate(pieces):
                if piece_bytes[i][d]:
                    segs.append((data.data_ptr() + piece_base[i][d], send_buf.data_ptr() + pos, piece_bytes[i][d]))
                pos += piece_bytes[i][d]
        dev.copy_segments(segs)
        recv_buf = all_to_all_rows(dist, send_buf, send_splits, recv_splits, 1, group)
        # unpack: per piece, the slices of every source end to end - one launch
        got = [dev.empty(sum(pb), dev.torch.uint8) for pb in recv_piece_bytes]
        segs, pos = [], 0
        filled = [0] * len(got)
        for s_rank in range(world):
            for i, pb in enumerate(recv_piece_bytes):
                if pb[s_rank]:
                    segs.append((recv_buf.data_ptr() + pos, got[i].data_ptr() + filled[i], pb[s_rank]))
                pos += pb[s_rank]
                filled[i] += pb[s_rank]
        dev.copy_segments(segs)
        out_cols, out_extras, i = [], [], 0
        for tag, arg in layout:
            if tag == "str":
                out_cols.append(dev.string_col(got[i], got[i + 1], n_in))
                i += 2
            elif tag == "fixed":
                out_cols.append(DCol(arg, got[i].view(dev.torch_dtype(arg)), n_in))
                i += 1
            else:
                out_extras.append(got[i].view(dev.torch_dtype(arg)))
                i += 1
        received = DBatch(list(batch.schema), out_cols, n_in, [0, n_in], total_units=batch.total_units)
        received.partitioned = True
        return received, out_extras

    @staticmethod
    def _piece_tags(layout: Sequence[tuple[str, Any]]) -> list[tuple[str, Any]]:
        """Pieces of the exchange buffer in order: ("rows", bytes per row) or ("payload", None) for string bytes."""
        from . import hipspark as hs  # noqa: PLC0415

        out: list[tuple[str, Any]] = []
        for tag, arg in layout:
            if tag == "str":
                out += [("rows", 1), ("payload", None)]
            else:
                out.append(("rows", hs.KIND_BYTES[arg]))
        return out

    def _exchange_by_key(self, batch: Any, key_index: int) -> tuple[Any, Any]:
        """Route rows to the rank owning hash(key) % SHUFFLE_PARTITIONS (owner = partition % world); the
        partition id of every row travels along.  -> (received batch, received partition ids)."""
        batch = self.dev.resolve(batch)
        part = self.dev.partition_ids(batch, key_index, constants.SHUFFLE_PARTITIONS)
        dest = part % self.world  # u8 arithmetic on a per-row id: plumbing
        received, (rpart,) = self._exchange_rows(batch, dest.contiguous(), [part])
        return received, rpart

    def _exchange_partial_rows(self, batch: Any) -> Any:
        """Partial-aggregate rows -> the rank owning their key's partition (generic form of the shuffle between
        the two aggregation phases: any key type, any cardinality).  The global unit id of every row travels
        along so the final merge keeps the reference's order."""
        import torch  # noqa: PLC0415

        batch = self.dev.resolve(batch)
        order = batch.order if batch.order is not None else torch.zeros(batch.nrows, dtype=torch.int64,
                                                                         device=self.dev.device)
        part = self.dev.partition_ids(batch, 0, constants.SHUFFLE_PARTITIONS)
        received, (rorder,) = self._exchange_rows(batch, (part % self.world).contiguous(), [order[: batch.nrows].contiguous()])
        received.order = rorder
        return received

    def _gather_to_root(self, batch: Any) -> Any:
        """Rows that are spread over the ranks travel to rank 0, which writes the result."""
        import torch  # noqa: PLC0415

        batch = self._materialise(self.dev.resolve(batch), [])
        dest = torch.zeros(batch.nrows, dtype=torch.uint8, device=self.dev.device)
        received, _ = self._exchange_rows(batch, dest, [])
        received.partitioned = False
        return received

    # ---- consumers -------------------------------------------------------------------------------------
    def _materialise(self, batch: Any, pending: Sequence[Any]) -> Any:
        """Apply deferred WHERE conditions: compaction to a row list, then gather every column."""
        if not pending:
            return batch
        import numpy as np  # noqa: PLC0415

        from . import hipspark as hs  # noqa: PLC0415
        from .device import DCol  # noqa: PLC0415

        batch = self.dev.resolve(batch)
        sel, count = self.dev.filter_select(batch, pending)
        bounds = self.dev.to_device(np.asarray(batch.unit_rows, dtype=np.int64))
        unit_rows = [int(v) for v in self.dev.lower_bound(sel, count, bounds).tolist()]
        out = self.dev.gather_batch(batch, sel, count, unit_rows)
        out.unit_ids, out.total_units, out.partitioned = batch.unit_ids, batch.total_units, batch.partitioned
        if batch.order is not None:
            out.order = self.dev.gather_col(DCol(hs.I64, batch.order, batch.nrows), sel, count).data
        return out

    def _project_keeping_filters(self, batch: Any, pending: Sequence[Any], task: Any) -> Any:
        """WHERE -> SELECT -> GROUP BY without the compaction in between (BASELINE config 5): when every selected
        column is a plain column or a concatenation of dictionary-coded columns (nothing that could raise on a row
        the WHERE drops), the projection is computed over ALL rows - a pass over code bytes - and the conditions stay
        pending for the fused scan + aggregate kernel.  The columns they name ride along behind the selected ones.
        None = not applicable (the caller filters, gathers and projects as usual)."""
        from .device import DBatch  # noqa: PLC0415
        from .lowering import ProgramBuilder, unalias  # noqa: PLC0415

        if batch.lazy:
            return None
        helper = ProgramBuilder(batch.schema, batch.kinds, batch.dicts)
        out_cols: list[Any] = []
        for col in task.columns:
            bare = unalias(col)
            if _cls(bare) in ("Col", "SchemaCol"):
                out_cols.append(batch.cols[batch.column_index(bare.name)])
            elif helper.string_tag(bare):
                coded = self.dev.dict_concat(batch, helper.string_parts(bare), batch.nrows)
                if coded is None:
                    return None
                out_cols.append(coded)
            else:
                return None  # arithmetic may divide by zero on rows the WHERE would have dropped
        schema = list(task.inferred_schema)
        names = {name: i for i, (name, _) in enumerate(schema)}
        for cond in pending:
            for name in _plain_names(cond):
                src = batch.cols[batch.column_index(name)]
                if name in names:
                    if out_cols[names[name]] is not src:
                        return None  # the SELECT re-uses the name for something else
                    continue
                names[name] = len(schema)
                schema.append(batch.schema[batch.column_index(name)])
                out_cols.append(src)
        return DBatch(schema, out_cols, batch.nrows, list(batch.unit_rows) if batch.unit_rows else None, None,
                      unit_ids=batch.unit_ids, total_units=batch.total_units, order=batch.order,
                      partitioned=batch.partitioned)

    def _project(self, batch: Any, pending: Sequence[Any], task: Any) -> Any:
        from . import hipspark as hs  # noqa: PLC0415
        from .device import DBatch  # noqa: PLC0415
        from .lowering import ProgramBuilder, unalias  # noqa: PLC0415

        batch = self._materialise(batch, pending)
        helper = ProgramBuilder(batch.schema, batch.kinds)
        if any(helper.string_tag(unalias(c)) and _cls(unalias(c)) not in ("Col", "SchemaCol") for c in task.columns):
            batch = self.dev.resolve(batch)  # string concatenation sizes its output from the exact row count
        n = batch.nrows
        out_cols: list[Any] = [None] * len(task.columns)
        numeric: list[tuple[int, Any]] = []
        for i, col in enumerate(task.columns):
            bare = unalias(col)
            if _cls(bare) in ("Col", "SchemaCol"):
                out_cols[i] = batch.cols[batch.column_index(bare.name)]  # pass-through keeps the stored values
            elif helper.string_tag(bare):
                out_cols[i] = self.dev.concat_strings(batch, helper.string_parts(bare), n)
            else:
                numeric.append((i, col))
        if numeric:
            evaluated = self.dev.eval_numeric(batch, [c for _, c in numeric])
            for (i, _), (dcol, tag) in zip(numeric, evaluated):
                if tag == "B":
                    raise AssertionError("a comparison cannot be selected as a column (the reference has no BOOL type)")
                out_cols[i] = dcol
        return DBatch(list(task.inferred_schema), out_cols, n, list(batch.unit_rows) if batch.unit_rows else None,
                      batch.nrows_dev, unit_ids=batch.unit_ids, total_units=batch.total_units, order=batch.order,
                      partitioned=batch.partitioned)

    # ---- writers ---------------------------------------------------------------------------------------
    def _quantise_batch(self, batch: Any, schema: Schema) -> Any:
        from .device import DBatch  # noqa: PLC0415

        if len(schema) != len(batch.cols):
            raise ExecutionError(f"writer schema {schema} does not match batch {batch.schema}")
        cols = self.dev.quantise_cols(batch.cols, [t for _, t in schema], batch.n_dev_ptr)
        unchanged = all(a is b for a, b in zip(cols, batch.cols))
        return DBatch(list(schema), cols, batch.nrows, list(batch.unit_rows) if batch.unit_rows else None,
                      batch.nrows_dev, unit_ids=batch.unit_ids, total_units=batch.total_units, order=batch.order,
                      slab=batch.slab if unchanged else None, slab_layout=batch.slab_layout if unchanged else None,
                      slab_cols=batch.slab_cols)

    def _write_result(self, batch: Any, schema: Schema, stage_id: str) -> JobResult:
        from . import hipspark as hs  # noqa: PLC0415
        from .device import RetryWithLargerDictionary  # noqa: PLC0415

        quantised = self._quantise_batch(batch, schema)
        raw, nrows, flags = self.dev.download_batch(quantised, schema, self._remote_flags)  # the one host round trip
        self._remote_flags = None
        if self.dev.rec is not None:
            self.dev.rec.result = (list(schema), stage_id)
        if self.dist is not None and getattr(self, "_generic_exchange_used", False):
            from .distributed import or_flags  # noqa: PLC0415

            flags = or_flags(self.dist, flags, self.dev.device, self.group)  # same decision on every rank
            self._generic_exchange_used = False
        if os.environ.get("HIPSPARK_DEBUG_FLAGS"):
            print(f"[hipspark] result flags {flags:#x} nrows {nrows} caps {self._caps}", flush=True)
        self._act_on_flags(flags)
        return self._emit_result(raw, nrows, schema, stage_id)

    def _act_on_flags(self, flags: int) -> None:
        """The one place a query's status word is turned into a decision - every result path (general write, short
        tail, streamed ranges) ends here, so no bit is handled on one path and ignored on another:
        restart without a fused path, retry with larger dictionaries, or the exception the reference raises."""
        from . import hipspark as hs  # noqa: PLC0415
        from .device import DeviceError, RetryWithLargerDictionary  # noqa: PLC0415

        if not flags:
            return
        if flags & hs.FLAG_ROUTE_STALE:
            # a sharded join build met other data than the split sizes it had agreed on (raised on every rank: the status
            # words meet before this decision): forget the sizes, run again - the next run agrees them anew
            self.dev.forget_routes()
            raise RestartQuery
        if flags & hs.FLAG_JOIN_DUP:
            # the in-place join met a build key twice: every join of this query takes the general path from now on
            self._no_fused_join.update(self._fused_join_tasks)
            raise RestartQuery
        if flags & hs.FLAG_MERGE_ROWS:
            # an on-chip merge ran on an upper bound of rows that did not fit and the real count was larger
            self._global_merge.update(self._lds_merges)
            raise RestartQuery
        if flags & (hs.FLAG_DICT_FULL | hs.FLAG_MERGE_FULL):
            raise RetryWithLargerDictionary(flags)
        self.dev.raise_for_flags(flags)
        if flags & ~hs.FLAG_KNOWN:
            raise DeviceError(f"status word {flags:#x} holds bits no result path handles")

    def _finish_short_tail(self, batch: Any, consumers: Sequence[Any], writer: Any, stage_id: str) -> JobResult:
        """[all-gather of the slabs] + ONE launch: final merge, projection, stored kinds, result image."""
        import torch  # noqa: PLC0415

        from . import hipspark as hs  # noqa: PLC0415
        from .device import RetryWithLargerDictionary, TierExceeded  # noqa: PLC0415

        tail = batch.tail
        merge = consumers[0]
        project = list(consumers[1].columns) if len(consumers) == 2 else None
        schema = writer.inferred_schema
        slab = tail["slab"]
        if tail.get("replicated"):
            # the fused join's unit tables were added up over the ranks before they were rounded into the slab: every
            # rank holds the same, complete partial rows
            gathered, world, n_order = slab, 1, max(tail["n_units"], 1)
        elif self.dist is not None:
            from .distributed import all_gather_slabs_into  # noqa: PLC0415

            gathered = torch.empty(self.world * tail["layout"].nbytes, dtype=torch.uint8, device=slab.device)
            timed = self.dev.exchange_events
            # the slab exchange follows the scan launch directly: when that launch is timed too, its end event IS the
            # exchange's begin (one event record less on the stream: each costs the next launch ~6 us)
            self.dev.exchange_begins_at_scan_end = timed is not None and self.dev.scan_events is not None
            if timed is not None and not self.dev.exchange_begins_at_scan_end:
                self.dev.op(timed[0].record)
            peers = self._peer_slabs(tail["layout"].nbytes)
            if peers is not None:
                # prototype (HIPSPARK_P2P_SLABS=1): stores into every peer's mapped buffer + device-side flags, no collective
                # (two library launches: recorded and replayed like every other launch of the query, arguments unchanged)
                peers.push(self.dev.stream, slab)
                peers.wait_into(self.dev.stream, tail["layout"].nbytes, gathered, self.dev.flags.data_ptr(),
                                recorded=self.dev.rec is not None)
                if self.dev.rec is not None:
                    self.dev.rec.keep.append((slab, gathered))
                self.p2p_exchanges += 1
            else:
                self.dev.op(all_gather_slabs_into, self.dist, slab, gathered, self.group)
            if timed is not None:
                self.dev.op(timed[1].record)
            world, n_order = self.world, batch.total_units
        else:
            gathered, world, n_order = slab, 1, max(tail["n_units"], 1)
        try:
            raw, nrows, flags = self.dev.aggregate_finish(tail, gathered, world, merge.agg_columns, merge.inferred_schema,
                                                          project, schema, self.merge_cap_hint, n_order,
                                                          cache_key=_uid(merge))
        except TierExceeded:
            # too many partial rows for one workgroup's LDS: this query takes the general path from now on
            self._no_short_tail.add(tail.get("task_id"))
            raise RestartQuery from None
        self.short_tails += 1
        if self.dev.rec is not None:
            self.dev.rec.result = (list(schema), stage_id)
        if self.dist is not None and getattr(self, "_generic_exchange_used", False):
            from .distributed import or_flags  # noqa: PLC0415

            flags = or_flags(self.dist, flags, self.dev.device, self.group)
            self._generic_exchange_used = False
        self._act_on_flags(flags)
        return self._emit_result(raw, nrows, schema, stage_id)

    def _peer_slabs(self, slab_bytes: int) -> Any:
        """The peer-to-peer slab exchange when it is switched on and the slab fits a slot (the same answer on every rank: the
        layout is), set up at the first such query - a collective moment, like the query itself."""
        if os.environ.get("HIPSPARK_P2P_SLABS", "0") != "1" or self.dist is None:
            return None
        if self._p2p is None:
            from .distributed import PeerSlabs, PeerSlabsUnavailable  # noqa: PLC0415

            try:
                # the library itself, never the recording proxy of whichever run happens to be first: PeerSlabs lives as long as
                # the engine and a stale proxy would keep appending its arguments to that old recording
                self._p2p = PeerSlabs(self.dist, self.group, self.rank, self.world, self.dev.device, self.dev._raw_lib)
            except PeerSlabsUnavailable as e:  # raised on every rank together: all of them keep the all-gather
                print(f"[hipspark] peer-to-peer slabs unavailable, using the all-gather: {e}", file=sys.stderr, flush=True)
                self._p2p = False
        if self._p2p is False:
            return None
        return self._p2p if self._p2p.fits(slab_bytes) and slab_bytes % 16 == 0 else None

    def _emit_result(self, raw: list, nrows: int, schema: Schema, stage_id: str) -> JobResult:
        self._job_seq += 1
        job_id = f"{self._job_prefix}-{self._job_seq}"
        if nrows == 0 or self.rank != 0:
            # empty result: the reference writes no file (tasks.py:405); multi-GPU: rank 0 owns the result
            return JobResult(job_id, self._executor_id, [])
        if self._result_root is None:
            # always a fresh sub-folder: __exit__ removes only what this engine created, never a directory the
            # caller handed in (which may hold other files - e.g. the input tables)
            base = Path(self._work_folder) if self._work_folder else constants.SHUFFLE_FOLDER
            self._result_root = base / f"hip-{uuid.uuid4().hex[:12]}"
            self._owned_dirs.add(self._result_root)
        out_file = self._result_paths.get(stage_id)
        if out_file is None:
            out_file = self._result_paths[stage_id] = self._result_root / str(stage_id) / "result.bin"
        # the rows are already in host memory: hand them over; the BlockFile is written when its path is read
        return JobResult(job_id, self._executor_id, [ResultFile(out_file, list(schema), raw, nrows)])
