"""Print the JIT source of BASELINE config 4's kernel (the shared-dictionary scan with the join's probe inside); no GPU
needed.  usage: jit_source_join.py > j.hip ; hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S
       --cuda-device-only -DHS_JIT_BUILD?  (the source includes hs_agg_kernel.h: add -Iinclude -Iminispark_amd/csrc)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minispark_amd import hipspark as hs
from minispark_amd.constants import ColumnType as T
from minispark_amd.lowering import lower_aggregate
from minispark_amd.sql import Col, Functions as F

lib = hs.load_library()
schema = [("o_orderpriority", T.STRING), ("l_quantity", T.FLOAT), ("l_extendedprice", T.FLOAT)]
kinds = [hs.STR, hs.F32, hs.F32]
dicts = [(b"1-URGENT", b"2-HIGH", b"3-MEDIUM", b"4-NOT SPECIFIED", b"5-LOW"), None, None]
low = lower_aggregate(schema, kinds, [], Col("o_orderpriority"),
                      [F.count().alias("n"), F.sum(Col("l_quantity")).alias("qty"), F.sum(Col("l_extendedprice")).alias("revenue"),
                       F.max(Col("l_extendedprice")).alias("max_price")], dicts)
n = len(low.program.columns)
cols = (hs.hs_col * (n + 1))()
for slot, ci in enumerate(low.program.columns):
    cols[slot].kind, cols[slot].fixed_len = (hs.JOIN8_CODE, 1) if kinds[ci] == hs.STR else (kinds[ci], -1)
cols[n].kind, cols[n].fixed_len = hs.JOIN8_UNIT, -1
prog, spec = low.program.to_struct(), low.spec()
src = C.create_string_buffer(1 << 16)
nb = C.c_int64(0)
rc = lib.hs_jit_compile_check_shared(cols, n + 1, low.key_slot, n, C.byref(prog), C.byref(spec), b"gfx950", C.byref(nb), src, len(src))
if rc:
    raise SystemExit(lib.hs_last_error().decode())
print(src.value.decode())
