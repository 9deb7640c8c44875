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
