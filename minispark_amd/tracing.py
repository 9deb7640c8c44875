"""Query tracing (reference: src/mini_spark/utils.py:83-135, a perfetto protobuf writer fed by TRACER.start / end
around stages and jobs, worker traces merged in ``save``).

Same call surface - ``new_track``, ``start``, ``end``, ``save`` - but the output is the Chrome Trace Event JSON
format, which https://ui.perfetto.dev opens directly and needs no protobuf dependency.  The HIP engine records a
span per query, per stage and per replayed launch sequence on the host track, and the scan kernel's duration
(HIP events) on a GPU track, when constructed with ``trace_file=...`` or run with ``HIPSPARK_TRACE=<file>``.

Round 2: EVERY launch of a traced query - scan, unit combine, finish, gathers, join build / probe,
partition ... also inside replayed recordings - appears as a slice on the GPU track (the library brackets its launches
with event pairs between hs_trace_begin / hs_trace_end, include/hipspark.h), and ``add_rocprof_kernel_trace`` merges a
rocprofv3 kernel-trace CSV as a further track.
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

    def add_rocprof_kernel_trace(self, csv_path: str | Path, track_name: str = "GPU (rocprofv3 --kernel-trace)",
                                 align_to_ns: int | None = None) -> int:
        """Merge a ``rocprofv3 --kernel-trace --output-format csv`` file as one more GPU track (the reference merges
        its workers' trace files the same way, utils.py:69-79).  rocprofv3 stamps kernels on its own clock: the
        first kernel is placed at ``align_to_ns`` (default: the start of the first slice already in this trace).
        -> number of kernels added."""
        import csv  # noqa: PLC0415

        rows = sorted(csv.DictReader(open(csv_path)), key=lambda r: int(r["Start_Timestamp"]))
        if not rows:
            return 0
        if align_to_ns is None:
            starts = [e["ts"] for e in self.events if e.get("ph") == "X"]
            align_to_ns = int(min(starts) * 1e3) if starts else time.time_ns()
        shift = align_to_ns - int(rows[0]["Start_Timestamp"])
        track = self.new_track(track_name)
        for r in rows:
            t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            self.complete(r["Kernel_Name"][:96], t0 + shift, t1 - t0, track, grid=r.get("Grid_Size_X", r.get("Grid_Size", "")),
                          workgroup=r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
        return len(rows)

    def save(self, filename: str | Path) -> None:
        Path(filename).write_text(json.dumps({"traceEvents": self.events, "displayTimeUnit": "ns"}))
