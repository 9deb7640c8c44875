"""The C-ABI library builds for gfx950, loads, and exports every symbol include/hipspark.h declares; the
run-time code generator translates and compiles the Q1 program for gfx950 without a GPU (hiprtc only
needs the compiler).  No compute calls here."""

from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

from minispark_amd import hipspark as hs
from tests.conftest import ROOT, load_golden


def _declared_functions() -> set[str]:
    text = (ROOT / "include" / "hipspark.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(hs_[a-z0-9_]+)\s*\(", text))


def test_every_declared_symbol_is_exported_and_bound():
    lib = hs.load_library()
    declared = _declared_functions()
    assert declared, "no declarations found in include/hipspark.h"
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, f"libhipspark.so does not export: {missing}"
    assert declared == set(hs.SIGNATURES), f"binding table differs from the header: {declared ^ set(hs.SIGNATURES)}"
    assert lib.hs_version() == 1
    assert hs.library_path().parent == ROOT / "minispark_amd"  # in-tree, not site-packages


def test_struct_layouts_match_the_header():
    assert C.sizeof(hs.hs_col) == 32
    assert C.sizeof(hs.hs_program) == 8 + 8 * hs.HS_MAX_INS + 8 * hs.HS_MAX_LIT + hs.HS_MAX_POOL
    assert C.sizeof(hs.hs_agg_spec) == 4 + 2 * hs.HS_MAX_ACC
    assert C.sizeof(hs.hs_agg_geom) == 40
    # and the sizes the library was compiled with
    lib = hs.load_library()
    mirrors = [hs.hs_col, hs.hs_program, hs.hs_agg_spec, hs.hs_agg_geom, hs.hs_chunk, hs.hs_slab_desc, hs.hs_finish_out,
               hs.hs_finish_spec, hs.hs_stage_plan, hs.hs_result_col, hs.hs_join8, hs.hs_join_stage_plan,
               hs.hs_select_stage_plan]
    for which, mirror in enumerate(mirrors):
        assert lib.hs_sizeof(which) == C.sizeof(mirror), mirror.__name__
    assert lib.hs_sizeof(len(mirrors)) == 0


def test_bad_arguments_are_refused_without_a_gpu():
    lib = hs.load_library()
    geom = hs.hs_agg_geom()
    units = (C.c_int64 * 2)(0, 100)
    assert lib.hs_agg_partial_geom(units, 1, 2, 3, C.byref(geom)) == 1  # group_cap must be a power of two
    assert b"bad arguments" in lib.hs_last_error()
    assert lib.hs_agg_partial_geom(units, 1, 16, 4096, C.byref(geom)) == 2  # does not fit in LDS
    assert lib.hs_agg_partial_geom(units, 1, 6, 4, C.byref(geom)) == 0
    assert geom.wg_threads == 256 and geom.chunk_rows % (256 * 4) == 0 and geom.n_chunks >= 1
    assert geom.lds_bytes == 4 * 16 + 4 * 6 * 256 * 8


def test_shared_tier_geometry_fits_its_rounds_of_workgroups():
    """The shared-dictionary tier runs equally long chunks, one 1024-lane workgroup each: a chunk more than the
    rounds of resident workgroups hold costs a whole extra round (29 blocks of 2 Mi rows once made 515 chunks
    for 512 places).  Chunks never span units and stay multiples of the 4096-row step."""
    lib = hs.load_library()
    geom = hs.hs_agg_geom()
    for n_units, rows in [(29, 2_097_152), (287, 2_097_152), (7, 1_000_003), (1, 50_000), (3, 12_345_678)]:
        bounds = [i * rows for i in range(n_units + 1)]
        units = (C.c_int64 * (n_units + 1))(*bounds)
        for n_acc, cap in [(2, 16), (2, 64), (2, 1024), (8, 256)]:
            assert lib.hs_agg_shared_geom(units, n_units, n_acc, cap, C.byref(geom)) == 0, lib.hs_last_error()
            resident = 2 if geom.lds_bytes * 2 + 2048 <= 160 * 1024 else 1
            assert geom.chunk_rows % 4096 == 0 and geom.wg_threads == 1024
            assert geom.n_chunks == sum(-(-(bounds[u + 1] - (bounds[u] & ~3)) // geom.chunk_rows) for u in range(n_units))
            if n_units < 512 * resident:
                assert geom.n_chunks <= 512 * resident, (n_units, rows, n_acc, cap, geom.n_chunks)
            assert geom.group_cap >= cap and geom.pad == 2 * geom.group_cap


def test_join_table_geometries_without_a_gpu():
    """The partitioned joins size their tables and workspaces on the host: the hashed form leaves ~1.75 slots per build row
    in whole windows (512 slots while 65 536 windows hold the build side, else 1024); the dense form refuses key ranges past
    2^29 slots."""
    lib = hs.load_library()
    for n in (0, 1, 585, 586, 1_000_000, 16 * 1024 * 1024, 38_000_000):
        slots = lib.hs_join_hash_slots(n)
        assert slots % 512 == 0 and slots >= max(512, n * 7 // 4), (n, slots)
        assert slots <= n * 7 // 4 + 1024
        assert lib.hs_join_hash_ws_bytes(n) >= 2 * n  # (at least the slot scratch)
    assert lib.hs_join_hash_slots(40_000_000) == 0 and lib.hs_join_hash_ws_bytes(40_000_000) == 0
    assert lib.hs_join_hash_slots(-1) == 0
    assert lib.hs_join_dense_ws_bytes(1000, 1 << 29) > 0
    assert lib.hs_join_dense_ws_bytes(1000, (1 << 29) + 1) == 0
    assert lib.hs_join_dense_aux_bytes(5) == 8 * 2 * 4 + 64


def test_jit_translates_and_compiles_q1_for_gfx950():
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.dataframe import DataFrame
    from minispark_amd.lowering import lower_aggregate
    from minispark_amd.plan import PhysicalPlan
    from minispark_amd.sql import Col, Functions, Lit
    from tests.queries import api_namespace, q1

    lib = hs.load_library()
    g = load_golden("q1_multiblock")
    api = api_namespace(lambda: DataFrame(engine=object()), Col, Functions, Lit)
    st = PhysicalPlan.generate_physical_plan(q1(api, g["paths"]["lineitem"]).task).stages[0]
    schema = st.producer.inferred_schema
    kind_of = {T.INTEGER: hs.I32, T.FLOAT: hs.F32, T.STRING: hs.STR, T.TIMESTAMP: hs.I64}
    kinds = [kind_of[t] for _, t in schema]
    low = lower_aggregate(schema, kinds, [st.consumers[0].condition], st.consumers[1].group_by_column,
                          st.consumers[1].agg_columns)
    cols = (hs.hs_col * len(low.program.columns))()
    for slot, ci in enumerate(low.program.columns):
        cols[slot].kind = kinds[ci]
        cols[slot].fixed_len = 1 if kinds[ci] == hs.STR else -1
    prog, spec = low.program.to_struct(), low.spec()
    src = C.create_string_buffer(32768)
    nbytes = C.c_int64(0)
    rc = lib.hs_jit_compile_check(cols, len(low.program.columns), low.key_slot, C.byref(prog), C.byref(spec), b"gfx950",
                                  C.byref(nbytes), src, len(src))
    assert rc == 0, (lib.hs_last_error(), lib.hs_jit_last_log()[:2000])
    assert nbytes.value > 4096
    text = src.value.decode()
    # six distinct accumulators, read - folded - written back together per row (one base address, constant offsets)
    assert "hs_agg_main_body<JitProg>" in text and text.count("= hs_acc_fold(") == 6 and "hs_f32x4" in text
    assert "cell[1280] = a5;" in text and "const unsigned long long k" in text  # 256-lane workgroup; hoisted literals


def test_expression_programs_translate_and_compile_for_gfx950_without_a_gpu():
    """hs_eval's compiled form: bytecode -> four-rows-per-lane kernel source -> hiprtc for gfx950."""
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.lowering import ProgramBuilder
    from minispark_amd.sql import Col, Lit

    lib = hs.load_library()
    schema = [("a", T.FLOAT), ("b", T.FLOAT), ("i", T.INTEGER), ("t", T.TIMESTAMP), ("s", T.STRING)]
    kinds = [hs.F32, hs.F32, hs.I32, hs.I64, hs.STR]
    b = ProgramBuilder(schema, kinds)
    exprs = [Col("a") * (Lit(1) - Col("b")), (Col("i") // 7 + Col("i") % 5) * 2, (Col("a") > 0.5) & (Col("t") <= "1998-09-02"),
             Col("s").like("%AIR%") | (Col("s") == "MAIL")]
    tags = [b.emit_out(o, e) for o, e in enumerate(exprs)]
    assert tags == ["F", "I", "B", "B"]
    prog = b.finish()
    cols = (hs.hs_col * len(prog.columns))()
    for slot, idx in enumerate(prog.columns):
        cols[slot].kind, cols[slot].fixed_len = kinds[idx], -1
    out_kinds = (C.c_int32 * 4)(hs.F64, hs.I64, hs.U8, hs.U8)
    pstruct = prog.to_struct()
    size = C.c_int64(0)
    src = C.create_string_buffer(1 << 16)
    rc = lib.hs_jit_compile_check_eval(cols, len(prog.columns), C.byref(pstruct), out_kinds, 4, b"gfx950", C.byref(size),
                                       src, len(src))
    assert rc == 0, (lib.hs_last_error(), lib.hs_jit_last_log())
    text = src.value.decode()
    assert size.value > 1000 and "k_eval_jit" in text and "hs_f32x4" in text and "hs_like_lit" in text
    assert "__builtin_nontemporal_store" in text  # f64 / i64 outputs leave with 16-byte stores


def test_shared_tier_programs_with_computed_units_and_dictionary_predicates_compile_without_a_gpu():
    """Round 2: the shared-dictionary kernel of the in-place join (per-row unit ids in the key word, runs of equal
    slots folded in registers before the LDS atomics) and a LIKE on a dictionary-coded column (HS_OP_DICTBIT)."""
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.lowering import lower_aggregate
    from minispark_amd.sql import Col, Functions as F

    lib = hs.load_library()
    schema = [("o_orderpriority", T.STRING), ("l_shipmode", T.STRING), ("l_quantity", T.FLOAT), ("l_extendedprice", T.FLOAT)]
    kinds = [hs.STR, hs.STR, hs.F32, hs.F32]
    dicts = [(b"1-URGENT", b"2-HIGH"), (b"AIR", b"MAIL", b"REG AIR"), None, None]
    low = lower_aggregate(schema, kinds, [Col("l_shipmode").like("%AIR%")], Col("o_orderpriority"),
                          [F.count().alias("n"), F.sum(Col("l_quantity")).alias("q"), F.max(Col("l_extendedprice")).alias("m")],
                          dicts)
    assert any((w & 0xFF) == hs.OP_DICTBIT for w in low.program.ins)
    assert low.program.lits[(low.program.ins[0] >> 32) & 0xFFFF] == 0b101  # AIR and REG AIR match, MAIL does not
    n = len(low.program.columns)
    cols = (hs.hs_col * (n + 1))()
    for slot, ci in enumerate(low.program.columns):
        cols[slot].kind, cols[slot].fixed_len = kinds[ci], (1 if kinds[ci] == hs.STR else -1)
    cols[n].kind, cols[n].fixed_len = hs.U8, -1  # the unit id column
    prog, spec = low.program.to_struct(), low.spec()
    src = C.create_string_buffer(1 << 16)
    nbytes = C.c_int64(0)
    for unit_col in (n, -1):
        rc = lib.hs_jit_compile_check_shared(cols, n + 1, low.key_slot, unit_col, C.byref(prog), C.byref(spec), b"gfx950",
                                             C.byref(nbytes), src, len(src))
        assert rc == 0, (lib.hs_last_error(), lib.hs_jit_last_log()[:3000])
        text = src.value.decode()
        assert "hs_agg_shared_body<JitProg>" in text and "hs_dictbit(" in text and "run_slot" in text
        assert ("hs_unit_key(" in text) == (unit_col >= 0)


def test_the_fused_probe_translates_and_compiles_for_gfx950_without_a_gpu():
    """Round 3: BASELINE config 4's kernel - the join's probe spliced into the shared-dictionary scan.  The GROUP BY key
    is a HS_JOIN8_CODE column, the unit a HS_JOIN8_UNIT column (both virtual: the probe side's key column looked up in
    the byte table while the keys are loaded); the interpreter kernels do not know them."""
    from minispark_amd.constants import ColumnType as T
    from minispark_amd.lowering import lower_aggregate
    from minispark_amd.sql import Col, Functions as F

    lib = hs.load_library()
    schema = [("o_orderpriority", T.STRING), ("l_quantity", T.FLOAT), ("l_extendedprice", T.FLOAT)]
    kinds = [hs.STR, hs.F32, hs.F32]
    dicts = [(b"1-URGENT", b"2-HIGH", b"3-MEDIUM"), None, None]
    low = lower_aggregate(schema, kinds, [Col("l_quantity") > 10], Col("o_orderpriority"),
                          [F.count().alias("n"), F.sum(Col("l_quantity")).alias("q"), F.max(Col("l_extendedprice")).alias("m")],
                          dicts)
    assert not low.program.code_columns
    n = len(low.program.columns)
    cols = (hs.hs_col * (n + 1))()
    for slot, ci in enumerate(low.program.columns):
        cols[slot].kind, cols[slot].fixed_len = (hs.JOIN8_CODE, 1) if kinds[ci] == hs.STR else (kinds[ci], -1)
    cols[n].kind, cols[n].fixed_len = hs.JOIN8_UNIT, -1
    prog, spec = low.program.to_struct(), low.spec()
    src = C.create_string_buffer(1 << 16)
    nbytes = C.c_int64(0)
    rc = lib.hs_jit_compile_check_shared(cols, n + 1, low.key_slot, n, C.byref(prog), C.byref(spec), b"gfx950",
                                         C.byref(nbytes), src, len(src))
    assert rc == 0, (lib.hs_last_error(), lib.hs_jit_last_log()[:3000])
    text = src.value.decode()
    assert "hs_agg_shared_body<JitProg>" in text and text.count("hs_join8_lookup(A.join") == 4
    assert text.count("hs_py_partition_inv(") == 4 and "hs_unit_key(" in text and "same ? vb[0] : x.jb[1]" in text
    # the loads ask, run() sorts out: nothing in load() depends on a byte it has just requested
    load_body = text[text.index("void load(const AggMainArgs"):text.index("void load_keys(")]
    assert "hs_py_partition" not in load_body and "vb[" not in load_body
    # outside the shared tier with that unit column the translator refuses (the caller then materialises the probe)
    rc = lib.hs_jit_compile_check_shared(cols, n + 1, low.key_slot, -1, C.byref(prog), C.byref(spec), b"gfx950",
                                         C.byref(nbytes), src, len(src))
    assert rc == 2 and b"HS_JOIN8" in lib.hs_last_error()
    assert lib.hs_join8_table_bytes(1) == hs.JOIN8_WINDOW and lib.hs_join8_table_bytes(hs.JOIN8_WINDOW + 1) == 2 * hs.JOIN8_WINDOW
    assert lib.hs_join8_ws_bytes(15_000_000, 60_000_000) > 4 * 15_000_000
    assert lib.hs_join8_build(None, None, None, 0, 0, None, 0, 1, None, None, None) == 1  # null arguments are refused


def test_stage_plan_blob_lowers_without_a_gpu_and_matches_the_library_mirror():
    """minispark_amd/stage.py: Q1 -> hs_stage_plan (what a cgo / JNI host would build); the ctypes mirrors of the
    stage-level structures have the sizes the library was compiled with."""
    from minispark_amd.dataframe import DataFrame
    from minispark_amd.sql import Col, Functions, Lit
    from minispark_amd.stage import lower_stage_plan
    from minispark_amd.workloads import api_namespace, q1

    lib = hs.load_library()
    assert lib.hs_sizeof(8) == C.sizeof(hs.hs_stage_plan) and lib.hs_sizeof(9) == C.sizeof(hs.hs_result_col)
    g = load_golden("q1_multiblock")
    api = api_namespace(lambda: DataFrame(engine=object()), Col, Functions, Lit)
    blob, path, schema = lower_stage_plan(q1(api, g["paths"]["lineitem"]).task)
    assert str(path) == g["paths"]["lineitem"] and blob.version == hs.HS_STAGE_PLAN_VERSION
    assert blob.n_cols == 6 and blob.spec.n_acc == 6 and blob.fin.n_fold == 6 and blob.fin.n_out == len(schema) == 9
    assert blob.fin_prog.n_ins > 0  # the three AVG = sum / count projections
    assert [blob.out_names[o].value.decode() for o in range(9)] == [n for n, _ in schema]
    assert blob.out_types[0] == 1 and blob.out_types[8] == 0  # STRING key ... INTEGER count


def test_join_stage_plan_blob_lowers_without_a_gpu():
    """minispark_amd/stage.py lower_join_stage_plan: BASELINE config 4's query -> hs_join_stage_plan (what a cgo / JNI host
    would build for the native JOIN stage); shapes the stage does not hold are refused on the host."""
    import pytest

    from minispark_amd.dataframe import DataFrame
    from minispark_amd.sql import Col, Functions, Lit
    from minispark_amd.stage import StageUnsupported, lower_join_stage_plan
    from minispark_amd.workloads import api_namespace, join_group

    lib = hs.load_library()
    assert lib.hs_sizeof(11) == C.sizeof(hs.hs_join_stage_plan)
    g = load_golden("join_group")
    api = api_namespace(lambda: DataFrame(engine=object()), Col, Functions, Lit)
    blob, build, probe, schema = lower_join_stage_plan(join_group(api, g["paths"]["orders"], g["paths"]["lineitem"]).task)
    assert (str(build), str(probe)) == (g["paths"]["orders"], g["paths"]["lineitem"])
    assert blob.version == hs.HS_JOIN_STAGE_PLAN_VERSION and blob.n_parts == 10
    assert (blob.build_key_col, blob.build_payload_col, blob.probe_key_col) == (0, 1, 0)
    assert blob.n_cols == 3 and list(blob.col_ids)[:3] == [-1, 1, 2] and blob.key_slot == 0  # the key is the build-side column
    assert blob.spec.n_acc == 4 and blob.fin.n_fold == 4 and blob.fin.n_out == len(schema) == 5
    assert [blob.out_names[o].value.decode() for o in range(5)] == ["o_orderpriority", "n", "qty", "revenue", "max_price"]
    C_, F = api.Col, api.F
    o = api.DataFrame().table(g["paths"]["orders"]).select(C_("o_orderkey"), C_("o_orderpriority"))
    li = api.DataFrame().table(g["paths"]["lineitem"]).select(C_("l_orderkey"), C_("l_quantity"))
    with pytest.raises(StageUnsupported, match="predicate on the build-side column"):
        lower_join_stage_plan(o.join(li, on=C_("o_orderkey") == C_("l_orderkey"), how="inner").filter(C_("o_orderpriority").like("1%"))
                              .group_by(C_("o_orderpriority")).agg(F.count()).task)
    with pytest.raises(StageUnsupported):
        lower_join_stage_plan(api.DataFrame().table(g["paths"]["lineitem"]).group_by(C_("l_orderkey")).agg(F.count()).task)


def test_radix_tier_plan_is_host_only_and_sizes_its_workspace():
    """hs_group_radix_plan (csrc/hs_radix.hip) touches no GPU: the workspace grows with the row count, a unit that fits
    one table needs no partition fan-out to speak of, unsupported key kinds and empty inputs are refused."""
    lib = hs.load_library()
    spec = hs.hs_agg_spec()
    spec.n_acc = 2
    spec.op[0], spec.op[1] = hs.AGG_SUM, hs.AGG_SUM
    spec.is_int[0], spec.is_int[1] = 0, 1
    kinds = (C.c_int32 * 2)(hs.F32, -1)  # SUM(f32 column), COUNT (a constant that does not travel)

    def ws(n, units, biggest, key_kind=hs.I32):
        plan = hs.hs_radix_plan()
        rc = lib.hs_group_radix_plan(key_kind, n, units, biggest, kinds, C.byref(spec), 1, C.byref(plan))
        return rc, (lib.hs_group_radix_ws_bytes(C.byref(plan)) if rc == 0 else 0)

    rc_small, small = ws(1000, 1, 1000)
    rc_mid, mid = ws(60_000_000, 29, 2_097_152)
    rc_big, big = ws(600_037_902, 287, 2_097_152)
    assert (rc_small, rc_mid, rc_big) == (0, 0, 0) and 0 < small < mid < big
    # two buffer sets of (4 B key + 4 B value) per row + provisional groups (8 B key + 2 x 4 B) + bookkeeping
    assert 32 * 60_000_000 <= mid <= 48 * 60_000_000
    assert ws(1000, 1, 1000, key_kind=hs.STR)[0] != 0 and ws(0, 1, 1)[0] != 0 and ws(10, 0, 10)[0] != 0
    assert b"hs_group_radix_plan" in lib.hs_last_error()


def test_scan_stage_lowering_inlines_a_projection_and_hands_over_a_computed_key():
    """Round 3: SELECT (l_orderkey % 331 - 100) AS bucket, ... GROUP BY bucket lowers to plan version 2 - the key as its own
    program over the table's columns, the other projected columns inlined; expressions that could raise on rows the WHERE
    drops, or that may not fit the stored INTEGER, are refused (the engine's general path takes them)."""
    import pytest

    from minispark_amd.dataframe import DataFrame
    from minispark_amd.io import BlockFile
    from minispark_amd.sql import Col, Functions as F, Lit
    from minispark_amd.stage import StageUnsupported, lower_stage_plan
    from tests.conftest import load_golden

    path = load_golden("many_groups")["paths"]["lineitem"]
    def frame(key):
        return (DataFrame(object()).table(path).filter(Col("l_shipdate") > "1992-03-01")
                .select(key.alias("bucket"), Col("l_extendedprice").alias("price"), (Col("l_tax") * 2).alias("t2"))
                .filter(Col("t2") < 0.15).group_by(Col("bucket")).agg(F.sum(Col("price") * (Lit(1) + Col("t2"))).alias("g"), F.count()))

    blob, table, schema = lower_stage_plan(frame(Col("l_orderkey") % 331 - 100).task)
    assert blob.version == 2 and blob.key_computed == 1 and blob.n_kcols == 1 and blob.col_ids[blob.key_slot] == -1
    names = [n for n, _ in BlockFile(path).file_schema]
    assert names[blob.kcol_ids[0]] == "l_orderkey"
    used = {names[blob.col_ids[i]] for i in range(blob.n_cols) if i != blob.key_slot}
    assert used == {"l_shipdate", "l_extendedprice", "l_tax"}  # projected names resolved to the table's columns
    assert [n for n, _ in schema][0] == "bucket" and Path(table) == Path(path)
    for bad in (Col("l_orderkey") // Col("l_orderkey"),  # may divide by zero on a row the WHERE drops
                Col("l_orderkey") * 4,                    # may not fit the stored INTEGER
                Col("l_orderkey") * Col("l_orderkey")):
        with pytest.raises(StageUnsupported):
            lower_stage_plan(frame(bad).task)


def test_no_kernel_of_the_library_lives_in_scratch_memory():
    """Register budget of the built library (tools/register_report.py reads the code objects' metadata, no GPU): the
    interpreter kernels of the fused aggregate - the fallback when hiprtc is unavailable - used to spill 440-712 bytes per
    lane at 1024 lanes; they now run 512 / 256 lanes wide and nothing spills.  The only
    kernels allowed a few spilled dwords are the radix partition's scatter kernels, which pin 8 waves per SIMD (two
    1024-lane workgroups per CU) on purpose."""
    from tools.register_report import kernels

    rows = kernels()
    assert len(rows) > 200
    spilling = {r["name"]: r["scratch"] for r in rows if r["scratch"]}
    assert all("k_rx_scatter" in name and size <= 64 for name, size in spilling.items()), spilling
    assert not any("k_agg" in name or "k_eval" in name or "k_join" in name for name in spilling)
