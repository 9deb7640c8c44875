"""Generated source + gfx950 ISA of the Q1 scan kernel (k_agg_jit) without a GPU:
    python tools/probes/dump_q1_isa.py /tmp/isa   ->  /tmp/isa/q1_jit.hip, /tmp/isa/q1_jit.s"""
import ctypes as C
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from minispark_amd import hipspark as hs  # noqa: E402
from minispark_amd.constants import ColumnType as T  # noqa: E402
from minispark_amd.dataframe import DataFrame  # noqa: E402
from minispark_amd.lowering import lower_aggregate  # noqa: E402
from minispark_amd.plan import PhysicalPlan  # noqa: E402
from minispark_amd.sql import Col, Functions, Lit  # noqa: E402
from tests.conftest import load_golden  # noqa: E402
from tests.queries import api_namespace, q1  # noqa: E402

out = Path(sys.argv[1] if len(sys.argv) > 1 else "/tmp/isa")
out.mkdir(parents=True, exist_ok=True)
lib = hs.load_library()
g = load_golden("q1_multiblock")
api = api_namespace(lambda: DataFrame(engine=object()), Col, Functions, Lit)
st = PhysicalPlan.generate_physical_plan(q1(api, g["paths"]["lineitem"]).task).stages[0]
schema = st.producer.inferred_schema
kind_of = {T.INTEGER: hs.I32, T.FLOAT: hs.F32, T.STRING: hs.STR, T.TIMESTAMP: hs.I64}
kinds = [kind_of[t] for _, t in schema]
low = lower_aggregate(schema, kinds, [st.consumers[0].condition], st.consumers[1].group_by_column, st.consumers[1].agg_columns)
cols = (hs.hs_col * len(low.program.columns))()
for slot, ci in enumerate(low.program.columns):
    cols[slot].kind = kinds[ci]
    cols[slot].fixed_len = 1 if kinds[ci] == hs.STR else -1
prog, spec = low.program.to_struct(), low.spec()
src = C.create_string_buffer(65536)
nbytes = C.c_int64(0)
rc = lib.hs_jit_compile_check(cols, len(low.program.columns), low.key_slot, C.byref(prog), C.byref(spec), b"gfx950", C.byref(nbytes), src, len(src))
assert rc == 0, lib.hs_last_error()
(out / "q1_jit.hip").write_text(src.value.decode())
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value",
                f"-I{ROOT / 'minispark_amd' / 'csrc'}", f"-I{ROOT / 'include'}", "-S", "--cuda-device-only", "-o", str(out / "q1_jit.s"),
                str(out / "q1_jit.hip")], check=True)
print(out / "q1_jit.s")
