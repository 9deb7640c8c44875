"""Boundary data types of the engine plug-in surface (reference: src/mini_spark/jobs.py:16-54).

``JobResult`` / ``OutputFile`` are what ``ExecutionEngine.execute_full_task`` returns and what
``collect_results`` consumes; ``ScanJob`` / ``LoadShuffleFilesJob`` / ``JoinJob`` name the three kinds of
work unit (one per file block, one per shuffle partition, one per join partition).  The HIP engine
does not serialise jobs to a worker process - a job is a row range (``unit``) of a device batch -
so there is no ``encode()`` wire format here.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path
from uuid import uuid4


@dataclass(frozen=True)
class OutputFile:
    file_path: Path
    partition: int = 0


class ResultFile:
    """The final result of a query, handed over in host memory: the rows arrived with the query's single
    device->host copy, so ``collect_results`` serves them directly.  It still stands for the BlockFile the
    reference's engines write (tasks.py:391-410): reading ``file_path`` writes that file on first use."""

    __slots__ = ("_path", "partition", "schema", "_raw", "nrows", "_written")

    def __init__(self, path: Path, schema: list, raw: list, nrows: int, partition: int = 0) -> None:
        self._path, self.partition, self.schema, self._raw, self.nrows = path, partition, schema, raw, nrows
        self._written = False

    @property
    def raw(self) -> list:
        from .io import LazyRaw  # noqa: PLC0415

        if isinstance(self._raw, LazyRaw):
            self._raw = self._raw.build()
        return self._raw

    @property
    def file_path(self) -> Path:
        if not self._written:
            from .constants import ROWS_PER_BLOCK  # noqa: PLC0415
            from .io import BlockFile, write_single_block_file  # noqa: PLC0415

            self._path.parent.mkdir(parents=True, exist_ok=True)
            if self.nrows <= ROWS_PER_BLOCK:
                write_single_block_file(self._path, self.schema, self.raw)
            else:
                BlockFile(self._path, list(self.schema)).write_raw(list(self.schema), self.raw)
            self._written = True
        return self._path

    def rows(self):  # noqa: ANN201
        from .io import LazyRaw, _row_builder, rows_list_from_raw  # noqa: PLC0415

        if isinstance(self._raw, LazyRaw):  # a small result: its values were decoded with the hand-over
            names = [n for n, _ in self.schema]
            return _row_builder(len(names))(names, self._raw.py_columns) if names else []
        return rows_list_from_raw(self.schema, self.raw)

    def columns(self) -> dict:
        """The result column-wise, as it arrived from the device: name -> numpy array (INTEGER i4, FLOAT f4, TIMESTAMP i8
        microseconds) or list of str.  No per-row Python objects: what a caller with 10^5 result rows wants."""
        from .io import StrCol  # noqa: PLC0415

        return {name: (c.to_list() if isinstance(c, StrCol) else c) for (name, _), c in zip(self.schema, self.raw)}


@dataclass
class JobResult:
    job_id: str
    executor_id: str
    output_files: list[OutputFile]


@dataclass
class Job:
    id: str = field(default_factory=lambda: str(uuid4()))


@dataclass
class ScanJob(Job):
    file_path: Path = Path()
    block_id: int = 0


@dataclass
class LoadShuffleFilesJob(Job):
    partition: int = 0


@dataclass
class JoinJob(Job):
    partition: int = 0
