"""Print the JIT source the library generates for the Q1 scan (no GPU needed; hiprtc cross-compiles for gfx950).
usage: jit_source.py > q1.hip ; hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only
       -Iinclude -Iminispark_amd/csrc q1.hip"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minispark_amd import hipspark as hs
from minispark_amd.constants import ColumnType as T
from minispark_amd.dataframe import DataFrame
from minispark_amd.lowering import lower_aggregate
from minispark_amd.plan import PhysicalPlan
from minispark_amd.sql import Col, Functions, Lit
from tests.conftest import load_golden
from minispark_amd.workloads import api_namespace, q1

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
nb = C.c_int64(0)
rc = lib.hs_jit_compile_check(cols, len(low.program.columns), low.key_slot, C.byref(prog), C.byref(spec), b"gfx950", C.byref(nb), src, len(src))
if rc:
    raise SystemExit(lib.hs_last_error().decode())
print(src.value.decode())
