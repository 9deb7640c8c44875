"""minispark_amd - an MI355X-native execution engine for minispark-style queries.

One data-parallel hot path (BlockFile scan -> WHERE / projection -> hash-partition shuffle -> hash
group-by/aggregate -> hash join) as hand-written HIP kernels for gfx950 behind a C ABI
(include/hipspark.h), driven by :class:`minispark_amd.execution.HipExecutionEngine`, which satisfies
the reference's ``ExecutionEngine`` plug-in surface.  See DESIGN.md.
"""

__all__ = ["constants", "io", "sql", "tasks", "plan", "jobs", "dataframe", "execution"]
