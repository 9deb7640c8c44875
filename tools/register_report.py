"""Registers, scratch and LDS of every kernel in the built libhipspark.so (no GPU needed):

    python tools/register_report.py [> profiles/rNN_kernel_registers.txt]

The shared object's gfx950 code objects are unbundled into a scratch folder (llvm-objdump --offloading) and their
AMDGPU metadata notes read (llvm-readelf --notes).  tests/test_abi.py holds the library to "no kernel spills to
scratch memory" with this report."""

from __future__ import annotations

import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")


def kernels(lib: Path | None = None) -> list[dict]:
    lib = lib or ROOT / "minispark_amd" / "libhipspark.so"
    out = []
    with tempfile.TemporaryDirectory(prefix="hs_regs_") as tmp:
        copy = Path(tmp) / lib.name
        shutil.copy(lib, copy)
        subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(copy)], check=True, capture_output=True, cwd=tmp)
        for code in sorted(Path(tmp).glob(f"{lib.name}.*gfx950*")):
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(code)], check=True, capture_output=True, text=True).stdout
            for block in notes.split("  - .agpr_count:")[1:]:
                def field(name: str) -> str:
                    m = re.search(rf"\.{name}:\s+(\S+)", block)
                    return m.group(1) if m else "0"

                out.append({"name": field("name").strip("'"), "vgpr": int(field("vgpr_count")), "sgpr": int(field("sgpr_count")),
                            "agpr": int(block.split()[0]), "scratch": int(field("private_segment_fixed_size")),
                            "lds": int(field("group_segment_fixed_size")), "max_wg": int(field("max_flat_workgroup_size"))})
    return sorted(out, key=lambda k: (-k["scratch"], -k["vgpr"], k["name"]))


if __name__ == "__main__":
    rows = kernels(Path(sys.argv[1]) if len(sys.argv) > 1 else None)
    print(f"{len(rows)} kernels in libhipspark.so; {sum(1 for r in rows if r['scratch'])} use scratch memory")
    print(f"{'scratch B':>9} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'static lds':>10} {'max wg':>6}  kernel")
    for r in rows:
        print(f"{r['scratch']:>9} {r['vgpr']:>5} {r['agpr']:>5} {r['sgpr']:>5} {r['lds']:>10} {r['max_wg']:>6}  {r['name']}")
