"""Query tracing (reference: src/mini_spark/utils.py:83-135, a perfetto protobuf writer fed by TRACER.start / end
around stages and jobs, worker traces merged in ``save``).

Same call surface - ``new_track``, ``start``, ``end``, ``save`` - but the output is the Chrome Trace Event JSON
format, which https://ui.perfetto.dev opens directly and needs no protobuf dependency.  The HIP engine records a
span per query, per stage and per replayed launch sequence on the host track, and the scan kernel's duration
(HIP events) on a GPU track, when constructed with ``trace_file=...`` or run with ``HIPSPARK_TRACE=<file>``.
"""

from __future__ import annotations

import json
import time
from pathlib import Path

MAIN_TRACK = 1


class Tracer:
    def __init__(self) -> None:
        self.events: list[dict] = []
        self.tracks: dict[int, str] = {}
        self._open: dict[int, list[tuple[str, int]]] = {}
        self.define_custom_track(MAIN_TRACK, "Main System")

    def new_track(self, name: str, parent_track_uuid: int = MAIN_TRACK) -> int:  # noqa: ARG002 - flat tracks
        uuid = 1000 + len(self.tracks)
        self.define_custom_track(uuid, name)
        return uuid

    def define_custom_track(self, track_uuid: int, name: str, parent_track_uuid: int | None = None) -> None:  # noqa: ARG002
        if track_uuid not in self.tracks:
            self.tracks[track_uuid] = name
            self.events.append({"ph": "M", "name": "thread_name", "pid": 1, "tid": track_uuid, "args": {"name": name}})

    def start(self, name: str, track_uuid: int = -1) -> None:
        track = MAIN_TRACK if track_uuid == -1 else track_uuid
        self._open.setdefault(track, []).append((name, time.time_ns()))

    def end(self, track_uuid: int = -1) -> None:
        track = MAIN_TRACK if track_uuid == -1 else track_uuid
        name, t0 = self._open[track].pop()
        self.complete(name, t0, time.time_ns() - t0, track)

    def complete(self, name: str, start_ns: int, duration_ns: int, track_uuid: int = MAIN_TRACK, **args: object) -> None:
        """A finished slice with explicit times (used for GPU kernel durations measured with events)."""
        ev = {"ph": "X", "name": name, "pid": 1, "tid": track_uuid, "ts": start_ns / 1e3, "dur": max(duration_ns, 0) / 1e3}
        if args:
            ev["args"] = args
        self.events.append(ev)

    def save(self, filename: str | Path) -> None:
        Path(filename).write_text(json.dumps({"traceEvents": self.events, "displayTimeUnit": "ns"}))
