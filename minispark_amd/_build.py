"""Build libhipspark.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

``python -m minispark_amd._build`` or ``build()``; called by ``__graft_entry__.build()``.  The shared
object lands next to this file (git-ignored, but it travels to the GPU box with the snapshot).
"""

from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libhipspark.so"
STAMP = PKG / ".libhipspark.stamp"
SOURCES = ["hs_agg.hip", "hs_ops.hip"]
HEADERS = [CSRC / "hs_device.h", PKG.parent / "include" / "hipspark.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-value"]


def _digest() -> str:
    h = hashlib.sha256()
    for p in [*(CSRC / s for s in SOURCES), *HEADERS]:
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP operator library cannot be built")


def build(force: bool = False, verbose: bool = True) -> Path:
    digest = _digest()
    if not force and LIB.exists() and STAMP.exists() and STAMP.read_text().strip() == digest:
        return LIB
    cmd = [hipcc_path(), *FLAGS, "-o", str(LIB), *[str(CSRC / s) for s in SOURCES]]
    if verbose:
        print("[minispark_amd] building", LIB.name, "for gfx950 ...", file=sys.stderr, flush=True)
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    STAMP.write_text(digest)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
