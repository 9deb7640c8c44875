/* hipspark.h - C ABI of libhipspark.so: the MI355X (gfx950) operator library behind
 * minispark_amd.execution.HipExecutionEngine.
 *
 * Every entry point replaces one operator loop of the reference (david-westreicher/minispark); the
 * reference file:line each one stands in for is cited on the declaration.  The reference's only
 * native boundary is a per-query generated Zig program fed jobs over stdin
 * (src/mini_spark/execution.py:182-219, zig-src/src/job.zig:3-57); this ABI is what a maintainer would
 * bind instead (ctypes stub: INTEGRATION.md).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no C++ / torch types cross the boundary.
 *  - All data pointers are DEVICE pointers unless a parameter is named host_*.  The caller owns all
 *    memory (inputs, outputs, workspaces); the library never allocates, never synchronises the
 *    device and never copies to the host, so every call can be captured into a hipGraph.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *  - Return value: 0 = launched ok, nonzero = HS_E_* (bad arguments; nothing was launched);
 *    hs_last_error() returns a thread-local message.  Data-dependent failures (division by zero,
 *    i32/f32 overflow at a quantisation point, dictionary capacity exceeded) are reported
 *    asynchronously in the caller-provided `flags` word(s), see HS_FLAG_*.
 *  - Rows of a batch are split into `units`: contiguous row ranges [unit_rows[u], unit_rows[u+1]).
 *    A unit is the reference's job: one BlockFile block for a scan (plan.py:90-93), one shuffle
 *    partition after a join (plan.py:99-109).  Partial aggregates are produced per unit because the
 *    reference quantises them to f32/i32 per job (tasks.py:373 -> io.py:87-94).
 */
#ifndef HIPSPARK_H
#define HIPSPARK_H

#ifndef HS_JIT_BUILD /* the run-time compiler (hiprtc) brings its own fixed-width types */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define HS_VERSION 1

/* ---- error codes (return values) ---- */
#define HS_OK 0
#define HS_E_ARG 1      /* malformed argument (null pointer, bad kind, bad program) */
#define HS_E_LIMIT 2    /* exceeds a compile-time limit (HS_MAX_*) */
#define HS_E_LAUNCH 3   /* hipLaunchKernel failed */

/* ---- asynchronous status bits written (atomically OR-ed) into a device `flags` word ---- */
#define HS_FLAG_DIV_ZERO 0x1u      /* reference: ZeroDivisionError raised by operator.truediv/floordiv/mod */
#define HS_FLAG_INT_OVERFLOW 0x2u  /* reference: OverflowError in int.to_bytes(4, signed=True), io.py:90 */
#define HS_FLAG_FLT_OVERFLOW 0x4u  /* reference: OverflowError in struct.pack('<f'), io.py:94 */
#define HS_FLAG_DICT_FULL 0x8u     /* more distinct group keys than the launch's capacity: retry larger */
#define HS_FLAG_BAD_PROGRAM 0x10u  /* interpreter met an op it cannot run in this kernel */
#define HS_FLAG_TYPE_ASSERT 0x40u   /* a FLOAT MIN/MAX never left its int identity (MAX_INT / MIN_INT): the reference
                                     * then fails `assert type(val) is float` at the file write, io.py:93 */
#define HS_FLAG_STR_TOO_LONG 0x20u /* a concatenated string exceeds 255 bytes (BlockFile length byte, io.py:18) */
#define HS_FLAG_JOIN_DUP 0x80u     /* hs_join_build_unique met a key twice: the caller takes the general (CSR) join */
#define HS_FLAG_MERGE_ROWS 0x200u /* hs_agg_merge_small was given an upper bound of rows that does not fit LDS and
                                    ran with what fits; the device-side count turned out larger: use the HBM-tier merge */
#define HS_FLAG_ROUTE_STALE 0x800u /* hs_join8_route: the build rows no longer route the way the cached split sizes say */
#define HS_FLAG_PEER_TIMEOUT 0x400u /* hs_slab_wait: a peer's slab did not arrive within the time limit (peer-to-peer exchange) */
#define HS_FLAG_MERGE_FULL 0x100u /* the FINAL merge met more distinct keys than merge_cap (hs_agg_merge_small /
                                    hs_agg_finish): grow that capacity, the per-unit one (HS_FLAG_DICT_FULL) is fine */

/* ---- storage kinds of a device column ---- */
#define HS_I32 0 /* INTEGER as stored in a BlockFile */
#define HS_F32 1 /* FLOAT as stored in a BlockFile */
#define HS_I64 2 /* TIMESTAMP (us) as stored; also in-flight INTEGER */
#define HS_F64 3 /* in-flight FLOAT (the reference computes in Python floats) */
#define HS_STR 4 /* STRING: lens[nrows] + payload bytes + (offs[nrows+1] unless fixed_len >= 0) */
#define HS_U8 5  /* boolean mask */
/* Virtual columns of the fused join + aggregate (hs_agg_shared_join8, round 3): `data` = the probe side's INTEGER key
 * column; the value of row i is looked up in the hs_join8 byte table while the aggregate scans - never stored. */
#define HS_JOIN8_CODE 16 /* the table byte of the row's key: the build side's 1-byte payload (a dictionary code) */
#define HS_JOIN8_UNIT 17 /* python_hash(key) % n_parts (the row's shuffle partition), 0xff = the key has no match */
/* the byte table those lookups read (built by hs_join8_build, see the join section below) */
typedef struct hs_join8 {
    const uint8_t* table; /* hs_join8_table_bytes(slots) bytes */
    int64_t slots;        /* key range covered: keys key_min .. key_min + slots - 1 */
    int32_t key_min;
    int32_t n_parts;      /* shuffle partitions (the reference's SHUFFLE_PARTITIONS): unit = python_hash(key) % n_parts */
} hs_join8;

typedef struct hs_col {
    int32_t kind;        /* HS_I32 ... HS_U8 */
    int32_t fixed_len;   /* HS_STR: byte length shared by every row, or -1 */
    const void* data;    /* values, or the concatenated string payload */
    const uint8_t* lens; /* HS_STR: per-row byte length */
    const int64_t* offs; /* HS_STR with fixed_len < 0: exclusive prefix sum of lens, [nrows+1] */
} hs_col;

/* ---- expression bytecode ------------------------------------------------------------------------
 * A stack program over 64-bit cells (f64 or i64; booleans are i64 0/1), produced by
 * minispark_amd/lowering.py from the reference's expression trees (sql.py:241-303: operator set
 * + - * / // % == != < <= > >= & |, INT op FLOAT -> FLOAT, '/' always FLOAT; sql.py:166-212 LIKE).
 * One instruction = 8 bytes: op:8 | sp:8 (stack depth BEFORE the instruction) | a:16 | b:16 | c:16.
 */
#define HS_MAX_INS 96
#define HS_MAX_LIT 32
#define HS_MAX_POOL 256
#define HS_MAX_COLS 12
#define HS_MAX_ACC 16
#define HS_MAX_OUTS 16 /* output columns of one hs_eval / projection program */
#define HS_MAX_STACK 8

enum hs_op {
    HS_OP_END = 0,
    HS_OP_LD = 1,       /* a = column slot; push its row value widened to f64 / i64 */
    HS_OP_LIT = 2,      /* a = literal index; push lit[a] */
    HS_OP_ADD_F = 3, HS_OP_SUB_F = 4, HS_OP_MUL_F = 5, HS_OP_DIV_F = 6, HS_OP_FLOORDIV_F = 7, HS_OP_MOD_F = 8,
    HS_OP_ADD_I = 9, HS_OP_SUB_I = 10, HS_OP_MUL_I = 11, HS_OP_FLOORDIV_I = 12, HS_OP_MOD_I = 13,
    HS_OP_LT_F = 14, HS_OP_LE_F = 15, HS_OP_GT_F = 16, HS_OP_GE_F = 17, HS_OP_EQ_F = 18, HS_OP_NE_F = 19,
    HS_OP_LT_I = 20, HS_OP_LE_I = 21, HS_OP_GT_I = 22, HS_OP_GE_I = 23, HS_OP_EQ_I = 24, HS_OP_NE_I = 25,
    HS_OP_AND = 26, HS_OP_OR = 27,
    HS_OP_I2F = 28,     /* a = 0: convert top of stack i64->f64, a = 1: the cell below it */
    HS_OP_STRCMP_LIT = 29, /* a = column slot, b = literal index of (pool_off<<32|len), c = cmp (0 lt,1 le,2 gt,3 ge,4 eq,5 ne); push bool */
    HS_OP_STRCMP_COL = 30, /* a, b = column slots, c = cmp; push bool */
    HS_OP_LIKE = 31,       /* a = column slot, b = literal index of (pool_off<<32|len) of the LIKE pattern; push bool */
    HS_OP_FILTER = 32,  /* pop; the row survives iff nonzero (tasks.py:167-177) */
    HS_OP_AGG = 33,     /* pop; fold into accumulator a (tasks.py:295-310) */
    HS_OP_OUT = 34,     /* pop; store as output a (hs_eval) */
    HS_OP_KEY = 35,     /* resolve the GROUP BY slot of the surviving rows (between filters and AGGs) */
    HS_OP_DICTBIT = 36  /* a = slot of a dictionary-coded column (one code byte per row), b = index of the first of
                           c (1..4) literal words holding one bit per dictionary entry; push bit[code]: LIKE and
                           string comparisons with a literal, evaluated once per dictionary entry on the host */
};

typedef struct hs_program {
    uint32_t n_ins;
    uint32_t n_lit;
    uint64_t ins[HS_MAX_INS];
    uint64_t lit[HS_MAX_LIT];
    uint8_t pool[HS_MAX_POOL]; /* string literals / LIKE patterns */
} hs_program;

/* ---- aggregate descriptors ---- */
#define HS_AGG_SUM 0 /* acc = acc + x from 0           (tasks.py:299-302) */
#define HS_AGG_MIN 1 /* acc = min(acc, x) from MAX_INT (tasks.py:303-306) */
#define HS_AGG_MAX 2 /* acc = max(acc, x) from MIN_INT (tasks.py:307-310) */

typedef struct hs_agg_spec {
    int32_t n_acc;
    uint8_t op[HS_MAX_ACC];     /* HS_AGG_* */
    uint8_t is_int[HS_MAX_ACC]; /* 1: i64 accumulator (INTEGER aggregate), 0: f64 */
} hs_agg_spec;

const char* hs_last_error(void);
int hs_version(void);
/* sizeof() of ABI structure `which` as compiled: 0 hs_col, 1 hs_program, 2 hs_agg_spec, 3 hs_agg_geom, 4 hs_chunk,
 * 5 hs_slab_desc, 6 hs_finish_out, 7 hs_finish_spec, 8 hs_stage_plan, 9 hs_result_col, 10 hs_join8, 11 hs_join_stage_plan, 12 hs_select_stage_plan
 * (0 for anything else) - lets a
 * binding verify its mirror. */
size_t hs_sizeof(int32_t which);

/* =================================================================================================
 * A1  BlockFile decode helpers (reference io.py:112-153: STRING = nrows length bytes + payload)
 * ===============================================================================================*/

/* Exclusive prefix sum of u8 lengths -> i64 offsets[nrows+1]; also min/max length (for fixed_len
 * detection) into minmax[2].  ws: hs_scan_ws_bytes(nrows) bytes. */
size_t hs_scan_ws_bytes(int64_t nrows);
int hs_str_offsets(void* stream, const uint8_t* lens, int64_t nrows, int64_t* offs, int32_t* minmax, void* ws);

/* =================================================================================================
 * A4  Expression evaluation (reference tasks.py:32-35 project_column + sql.py:262-266 execute_row)
 * ===============================================================================================*/

/* Everywhere below, a `*_dev` row count (device int64, may be NULL) caps the host-side count: rows at or
 * beyond it are not touched.  It lets operators run on a batch whose size is only known on the device.
 *
 * Evaluate `prog` for every row; each HS_OP_OUT a stores into outs[a] with kind out_kinds[a]
 * (HS_F64, HS_I64 or HS_U8 for booleans).  If `sel` != NULL only rows sel[0..nrows) of the input
 * are evaluated (output row i <- input row sel[i]). */
int hs_eval(void* stream, const hs_col* cols, int32_t n_cols, const hs_program* prog, const int64_t* sel,
            int64_t nrows, const int64_t* nrows_dev, void* const* outs, const int32_t* out_kinds, int32_t n_outs,
            uint32_t* flags);

/* =================================================================================================
 * A3  Filter = stream compaction (reference tasks.py:167-177; zig task_utils.zig:9-51)
 * ===============================================================================================*/

/* mask[nrows] (u8, nonzero = keep) -> ascending row indices sel[0..*count).  Stable.
 * ws: hs_scan_ws_bytes(nrows). count is a device int64. */
int hs_compact(void* stream, const uint8_t* mask, int64_t nrows, int64_t* sel, int64_t* count, void* ws);

/* out[i] = src[idx[i]] for a fixed-width column of src_rows rows (elem_bytes in {1,2,4,8}).  Indices come from other
 * kernels; one outside [0, src_rows) reads as zero and raises HS_FLAG_BAD_PROGRAM in *flags instead of faulting. */
int hs_gather_fixed(void* stream, const void* src, int32_t elem_bytes, int64_t src_rows, const int64_t* idx, int64_t n,
                    const int64_t* n_dev, void* dst, uint32_t* flags);
/* STRING gather, two steps around an offsets scan: lengths first ... */
int hs_gather_str_lens(void* stream, const hs_col* src, int64_t src_rows, const int64_t* idx, int64_t n, uint8_t* out_lens,
                       uint32_t* flags);
/* (second pass, after hs_str_offsets over out_lens gave out_offs; rows refused by the first pass are skipped) */
int hs_gather_str_bytes(void* stream, const hs_col* src, int64_t src_rows, const int64_t* idx, int64_t n,
                        const int64_t* out_offs, uint8_t* out_data);

/* STRING '+' (reference sql.py:262-266 with operator.add on str; zig utils.zig:118-131).
 * parts: n_parts columns (HS_STR) or literals (kind = -1, data = host-copied into the launch: the
 * literal bytes pointer is a DEVICE pointer, fixed_len = its length).  Step 1 lengths, step 2 bytes. */
int hs_concat_lens(void* stream, const hs_col* parts, int32_t n_parts, int64_t nrows, uint8_t* out_lens,
                   uint32_t* flags);
int hs_concat_bytes(void* stream, const hs_col* parts, int32_t n_parts, int64_t nrows, const int64_t* out_offs,
                    uint8_t* out_data);

/* =================================================================================================
 * A5/A6  Partial hash aggregate per unit, fused with scan + WHERE + aggregate-argument evaluation
 *        (reference tasks.py:167-177 + 270-310 before_shuffle branch; quantisation io.py:87-94)
 * ===============================================================================================*/

/* Geometry of one hs_agg_partial launch (host-side helper, pure arithmetic). */
typedef struct hs_agg_geom {
    int32_t group_cap;   /* per-workgroup / per-unit dictionary capacity (power of two) */
    int32_t chunk_rows;  /* rows per workgroup */
    int32_t wg_threads;  /* lanes per workgroup: 256, 128 or 64 (each lane owns a private LDS table) */
    int32_t pad;         /* set by the *_geom helpers (shared tier: slots of a unit's table); pass it on unchanged */
    int64_t n_chunks;    /* grid size */
    size_t lds_bytes;    /* dynamic LDS of the main kernel */
    size_t ws_bytes;     /* workspace bytes */
} hs_agg_geom;

/* One workgroup's share of a unit: rows [row_begin, row_end) of unit `unit` (the first quad may start up
 * to 3 rows before the unit; those rows are masked by unit_begin). */
typedef struct hs_chunk {
    int64_t row_begin;   /* multiple of 4 */
    int64_t row_end;
    int64_t unit_begin;  /* first row of the unit */
    int64_t unit;
} hs_chunk;

/* host_unit_rows: HOST copy of the unit boundaries [n_units+1].  Returns HS_E_LIMIT when
 * group_cap * (n_acc+1) private tables do not fit in LDS (caller then uses hs_agg_partial_global). */
int hs_agg_partial_geom(const int64_t* host_unit_rows, int64_t n_units, int32_t n_acc, int32_t group_cap,
                        hs_agg_geom* out);
/* Fill the HOST arrays the launch needs on the device: chunks[geom->n_chunks] and unit_chunk0[n_units+1]
 * (first chunk of every unit).  Upload them once; they depend only on the unit boundaries and geometry. */
int hs_agg_partial_chunks(const int64_t* host_unit_rows, int64_t n_units, const hs_agg_geom* geom,
                          hs_chunk* host_chunks, int64_t* host_unit_chunk0);

/* prog: [filter ops ... HS_OP_FILTER]* then [value ops ... HS_OP_AGG a]* ; cols[key_col] is the
 * GROUP BY column.  chunks / unit_chunk0: device copies of what hs_agg_partial_chunks produced.  Outputs, all [n_units * group_cap] slot-major per unit:
 *   out_rep   : a row index holding the slot's key, or -1 for an empty slot
 *   out_acc   : [n_units * group_cap * n_acc] 64-bit cells, ALREADY QUANTISED like a shuffle-file
 *               write (FLOAT aggregates rounded to f32 and widened back, INTEGER range-checked)
 *   out_ngroups[n_units] : occupied slots per unit
 * ev_begin / ev_end: optional hipEvent_t recorded on `stream` immediately before / after the main
 * scan kernel (NULL = none) - how bench.py measures the kernel's duration live.
 * ws: geom->ws_bytes bytes, ZERO-FILLED BEFORE THE FIRST LAUNCH that uses it (it ends in one arrival counter per
 * unit: the workgroup that finishes a unit's last chunk combines the unit inside the scan kernel and leaves the
 * counter at zero again, so later launches re-use the buffer as it is).  HIPSPARK_FUSE_UNIT=0, or a unit without
 * rows, selects the separate combine launch (k_agg_unit) instead.
 */
int hs_agg_partial(void* stream, const hs_col* cols, int32_t n_cols, int32_t key_col, const hs_program* prog,
                   const hs_agg_spec* spec, const hs_chunk* chunks, const int64_t* unit_chunk0, int64_t n_units,
                   const hs_agg_geom* geom, int64_t* out_rep, uint64_t* out_acc, int32_t* out_ngroups, void* ws,
                   uint32_t* flags, void* ev_begin, void* ev_end);

/* Shared-dictionary tier of the same operator, for tens to thousands of groups per unit: one
 * 1024-lane workgroup per chunk keeps ONE table of geom->group_cap slots in LDS and updates it with LDS atomics;
 * chunk tables are merged into per-unit tables of geom->pad slots with global atomics; cells are then rounded like
 * hs_agg_partial's.  Outputs as hs_agg_partial with group_cap = geom->pad (feed them to hs_agg_pack with that
 * capacity): out_rep[n_units * pad], out_acc[n_units * pad * n_acc], out_ngroups[n_units]; ws: geom->ws_bytes.
 * Additions happen in hardware order: results are not bitwise reproducible run to run (within ~1e-13 relative). */
int hs_agg_shared_geom(const int64_t* host_unit_rows, int64_t n_units, int32_t n_acc, int32_t group_cap, hs_agg_geom* out);
int hs_agg_shared(void* stream, const hs_col* cols, int32_t n_cols, int32_t key_col, const hs_program* prog,
                  const hs_agg_spec* spec, const hs_chunk* chunks, int64_t n_units, const hs_agg_geom* geom,
                  int64_t* out_rep, uint64_t* out_acc, int32_t* out_ngroups, void* ws, uint32_t* flags, void* ev_begin,
                  void* ev_end);

/* The same with COMPUTED units: rows are not grouped by unit; cols[unit_col] (HS_U8, a preloaded slot < 8) gives
 * every row's unit id in [0, n_unit_tables), 0xff = the row takes no part.  This is how the probe side of a join is
 * aggregated per shuffle partition (the reference's JoinJob, plan.py:99-109) without ever being partitioned in
 * memory: hs_join_probe_unique writes the id column.  chunks / geom: from hs_agg_shared_geom over the ONE row range
 * [0, nrows); geom->pad = slots of ONE unit's table, which the caller may lower (power of two >= 16) to what a unit
 * is expected to hold - HS_FLAG_DICT_FULL reports a unit that outgrew it; outputs sized for n_unit_tables units of
 * geom->pad slots; ws: (n_unit_tables * geom->pad * 8 + 256, rounded up to 16) + geom->n_chunks * n_unit_tables *
 * geom->pad * n_acc * 8 bytes (every chunk leaves its cells in its own slice, a small kernel folds them per unit
 * cell: no contended global atomics).  The key must fit 56 bits of its key word (INTEGER, or a string of fixed length <= 6): HS_E_LIMIT else. */
int hs_agg_shared_units(void* stream, const hs_col* cols, int32_t n_cols, int32_t key_col, int32_t unit_col,
                        int32_t n_unit_tables, const hs_program* prog, const hs_agg_spec* spec, const hs_chunk* chunks,
                        const hs_agg_geom* geom, int64_t* out_rep, uint64_t* out_acc, int32_t* out_ngroups, void* ws,
                        uint32_t* flags, void* ev_begin, void* ev_end);

/* Round 3: hs_agg_shared_units with the join's probe INSIDE the scan.  cols[unit_col] is a
 * HS_JOIN8_UNIT column and the GROUP BY key may be a HS_JOIN8_CODE column (both virtual: `data` = the probe side's key
 * column, looked up in join->table per row; a lane's four consecutive keys share a lookup when equal); nothing is
 * written per row.  Needs the run-time compiler (HS_E_LIMIT without it: the caller materialises unit / payload bytes
 * with hs_join_probe_unique instead).  Leaves the units' tables RAW - key words unit_keys[n_unit_tables * geom->pad]
 * (HS_EMPTY = free slot; the unit id in the top byte) and un-rounded 64-bit cells unit_acc[n_unit_tables * geom->pad *
 * n_acc] - so that the shares of N ranks can be added up BEFORE the rounding the reference applies once per JoinJob
 * (tasks.py:373 -> io.py:87-94).  ws: geom->n_chunks * n_unit_tables * geom->pad * n_acc * 8 bytes; out_rep scratch
 * [n_unit_tables * geom->pad]. */
int hs_agg_shared_join8(void* stream, const hs_col* cols, int32_t n_cols, int32_t key_col, int32_t unit_col,
                        const hs_join8* join, int32_t n_unit_tables, const hs_program* prog, const hs_agg_spec* spec,
                        const hs_chunk* chunks, const hs_agg_geom* geom, int64_t* out_rep, uint64_t* unit_keys,
                        uint64_t* unit_acc, void* ws, uint32_t* flags, void* ev_begin, void* ev_end);
/* N ranks hold raw unit tables of the same shape over DIFFERENT rows of the same units (each rank probed its own
 * blocks): gathered = world x [16-byte header: flags u32, 12 pad][keys: slots x 8][cells: slots x n_acc x 8] with
 * slots = n_units * unit_cap.  Merges them in rank order (deterministic given the inputs) into out_keys / out_acc of
 * the same shape; header flags are OR-ed into *flags; a unit whose keys do not fit unit_cap raises HS_FLAG_DICT_FULL. */
int hs_agg_units_merge(void* stream, const uint8_t* gathered, int32_t world, int32_t n_units, int32_t unit_cap,
                       const hs_agg_spec* spec, uint64_t* out_keys, uint64_t* out_acc, uint32_t* flags);
/* Raw unit tables -> the exchange slab hs_agg_finish reads (hs_slab_desc below): row u * unit_cap + s <- slot s of unit
 * u: order key = u (or -1: free slot), key element from the low bytes of the key word, cells rounded to the stored
 * kinds exactly like a shuffle-file write (overflow / type flags into *flags).  desc->slab_rows >= n_units * unit_cap. */
struct hs_slab_desc;
int hs_agg_units_to_slab(void* stream, const uint64_t* unit_keys, const uint64_t* unit_acc, int32_t n_units,
                         int32_t unit_cap, const hs_agg_spec* spec, uint8_t* slab, const struct hs_slab_desc* desc,
                         uint32_t* flags);

/* Dense pack of the slot arrays: rows of unit u go to [pack_start[u], pack_start[u+1]).
 * out_cols[a] receives accumulator a as HS_F32 / HS_I32 storage (acc_kinds[a]) = the reference's
 * shuffle-file column; out_rep the representative rows (gather the key column with them).
 * unit_ids / out_unit (both optional): out_unit[row] = unit_ids[u] (or u), the id that orders the final
 * merge when partial rows of several GPUs meet (multi-GPU: the file block id). */
int hs_agg_pack(void* stream, const int64_t* rep, const uint64_t* acc, const int32_t* ngroups, int64_t n_units,
                int32_t group_cap, const hs_agg_spec* spec, int64_t* pack_start, int64_t* out_rep,
                void* const* out_cols, const int32_t* acc_kinds, const int64_t* unit_ids, int64_t* out_unit);

/* =================================================================================================
 * A7  Final merge of partial rows (reference tasks.py:290-292 after-shuffle branch)
 * ===============================================================================================*/

/* Input: a batch of partial rows: key column `key`, accumulator columns acc_cols[n_acc] (HS_F32 /
 * HS_I32 / HS_F64 / HS_I64).  Partials of a key are folded in fp64 / i64 in ascending (order[row], row)
 * order - with order == NULL in row order - which is the reference's order: block order of the shuffle
 * file.  Rows with order[row] < 0 are padding and ignored (fixed-size slabs exchanged between GPUs); valid
 * order keys lie in [0, n_order) and rows sharing a key are contiguous and ascending in the input (they
 * come from one producer).
 * Outputs are DENSE (group i = i-th occupied dictionary slot) and column-major: out_rep[i] = first input
 * row of group i, out_acc[a * cap + i] = un-rounded 64-bit cell of aggregate a, *out_ngroups = number of
 * groups (<= cap).  n_rows is an upper bound when n_rows_dev != NULL (then *n_rows_dev, a device int64, is
 * the exact count): lets a whole query run without a host round trip between its kernels.  Everything is staged in
 * LDS: when n_rows does not fit (HS_E_LIMIT without n_rows_dev) but the count is on the device, the merge runs with as
 * many rows as fit and raises HS_FLAG_MERGE_ROWS if *n_rows_dev is larger (the caller then takes the HBM-tier merge);
 * more distinct keys than `cap` raise HS_FLAG_MERGE_FULL. */
int hs_agg_merge(void* stream, const hs_col* key, const hs_col* acc_cols, const hs_agg_spec* spec,
                 const int64_t* order, int64_t n_order, int64_t n_rows, const int64_t* n_rows_dev, int32_t cap,
                 int64_t* out_rep, uint64_t* out_acc, int64_t* out_ngroups, uint32_t* flags);

/* =================================================================================================
 * A6/A9  Hash partitioning (reference tasks.py:347-375 WriteToShufflePartitions.write)
 * ===============================================================================================*/

/* part[i] = python_hash(key[i]) % n_parts for INTEGER / TIMESTAMP keys (hash(int) = int, -1 -> -2,
 * floor-mod: tasks.py:362); STRING keys use a fixed 64-bit FNV-1a (the reference's str hash is
 * seed-randomised, any choice is a valid outcome). */
int hs_partition_ids(void* stream, const hs_col* key, const int64_t* sel, int64_t nrows, int32_t n_parts,
                     uint8_t* part);
/* Stable counting sort of row indices by part id: perm[part_start[p] .. part_start[p+1]) = rows of
 * partition p in ascending row order.  part_start: device int64[n_parts+1].  ws: hs_partition_ws_bytes. */
size_t hs_partition_ws_bytes(int64_t nrows, int32_t n_parts);
int hs_partition_perm(void* stream, const uint8_t* part, int64_t nrows, int32_t n_parts, int64_t* perm,
                      int64_t* part_start, void* ws);

/* =================================================================================================
 * A8  Hash join (reference tasks.py:201-240 BroadcastHashJoinTask.generate_chunks)
 * ===============================================================================================*/

/* Build: open-addressing dictionary over the LEFT key column (table_cap = power of two >= 2*n_left):
 * table_keys[cap] (u64 key words), table_reps[cap] (a left row holding the slot's key; used to
 * compare long strings), plus CSR row lists: the left rows of slot s are
 * rows[slot_start[s] .. slot_start[s+1]) in ASCENDING row order - the reference appends row indices
 * in row order (tasks.py:216-217).  slot_start: [cap+1].  ws: hs_join_build_ws_bytes(). */
size_t hs_join_build_ws_bytes(int64_t n_left, int64_t table_cap);
int hs_join_build(void* stream, const hs_col* left_key, int64_t n_left, int64_t table_cap, uint64_t* table_keys,
                  int64_t* table_reps, int64_t* slot_start, int64_t* rows, void* ws, uint32_t* flags);
/* The same build over positions 0..n-1 of an input whose position i is row sel[i] (or row0 + i when sel is
 * NULL): positions[] receives POSITIONS, ascending inside every slot; table_reps[] holds row ids.
 * This is the global-memory tier of GROUP BY (any cardinality): see hs_group_fold. */
int hs_group_build(void* stream, const hs_col* key, const int64_t* sel, int64_t row0, int64_t n, int64_t table_cap,
                   uint64_t* table_keys, int64_t* table_reps, int64_t* slot_start, int64_t* positions, void* ws,
                   uint32_t* flags);
/* hs_group_build over EVERY unit of a batch in one pass (round 2: the HBM tier no longer loops over units on the
 * host): unit u owns positions [unit_bounds[u], unit_bounds[u + 1]) of the input and the table region
 * [region_base[u], region_base[u + 1]) - a power of two of slots, at least twice the unit's rows; table_cap =
 * region_base[n_units] (device arrays of n_units + 1 entries).  A key is looked up inside its unit's region only, so
 * the non-empty slots in ascending order are the groups in unit order. */
int hs_group_build_units(void* stream, const hs_col* key, const int64_t* sel, int64_t row0, int64_t n,
                         const int64_t* unit_bounds, const int64_t* region_base, int32_t n_units, int64_t table_cap,
                         uint64_t* table_keys, int64_t* table_reps, int64_t* slot_start, int64_t* positions, void* ws,
                         uint32_t* flags);
/* out[q] = how many elements of the ascending list sorted[0 .. n) are < queries[q] (device arrays; n_dev optional). */
int hs_lower_bound_i64(void* stream, const int64_t* sorted, int64_t n, const int64_t* n_dev, const int64_t* queries,
                       int64_t n_queries, int64_t* out);
/* ---- the HBM tier as a radix partition + on-chip ordered fold (csrc/hs_radix.hip; reference tasks.py:284-310) ----
 * GROUP BY of any cardinality over every unit of a batch; keys whose 64-bit key word is the key itself - INTEGER /
 * TIMESTAMP, FLOAT (also computed f64; 0.0 and -0.0 are one group, as in a Python dict), STRING columns of one fixed length
 * <= 7 bytes - or, as 2 .. 4 four-byte key word columns that are compared exactly, of one fixed length 8 .. 16 bytes with SUM / COUNT aggregates
 * over f32 / i32 columns (key_kind = HS_STR + 256 x length; HS_E_LIMIT for other strings): the (key, value...) tuples are moved
 * by one or two stable partition passes on hash bits of the key until a partition (a few hundred rows of one unit, still
 * in row order) fits a wave's LDS dictionary; one wave folds a partition's rows in order - the reference's sequential
 * fp64 / int sums, bit for bit - and the groups are made dense.  Three calls:
 *   hs_group_radix_plan   host only: fan-out, dictionary size and workspace layout from the row count n (> 0), the
 *                         number of units and the largest unit's row count.  val_kinds[a]: kind of aggregate a's value
 *                         column (HS_I32 / F32 / I64 / F64 / U8), or -1 for a constant argument (COUNT's 1);
 *   hs_group_radix_run    position i of the input is row sel[i] of `key` (row0 + i when sel is NULL); unit u owns
 *                         positions [unit_bounds[u], unit_bounds[u+1]) (device, [0] = 0, [n_units] = n).  val_cols[a]
 *                         are indexed by POSITION; data == NULL marks a constant whose accumulator cell is
 *                         const_cells[a].  quantise (plan): results as the shuffle file holds them (f32 / i32) or as
 *                         f64 / i64.  out_unit_groups[u] (device, n_units + 1) = groups before unit u, [n_units] = all.
 *                         A partition with more distinct keys than its dictionary raises HS_FLAG_DICT_FULL;
 *   hs_group_radix_emit   after the caller has read out_unit_groups[n_units]: dense key column (the key's kind) and one
 *                         array per aggregate (4 B f32 / i32 when quantising, else 8 B), groups in unit order. */
typedef struct hs_radix_plan {
    int64_t f[48]; /* opaque */
} hs_radix_plan;
int hs_group_radix_plan(int32_t key_kind, int64_t n, int32_t n_units, int64_t max_unit_rows, const int32_t* val_kinds,
                        const hs_agg_spec* spec, int32_t quantise, hs_radix_plan* plan);
size_t hs_group_radix_ws_bytes(const hs_radix_plan* plan);
int hs_group_radix_run(void* stream, const hs_radix_plan* plan, const hs_col* key, const int64_t* sel, int64_t row0,
                       const int64_t* unit_bounds, const hs_col* val_cols, const uint64_t* const_cells,
                       const hs_agg_spec* spec, void* ws, int64_t* out_unit_groups, uint32_t* flags);
int hs_group_radix_emit(void* stream, const hs_radix_plan* plan, void* ws, void* out_key, void* const* out_acc);
/* ---- the general inner join on INTEGER keys over a dense key range (round 4; csrc/hs_radix.hip) -------------------------
 * Reference: BroadcastHashJoinTask.generate_chunks tasks.py:201-240 (zig twin tasks.zig:70-194; duplicate keys multiply,
 * tasks.zig:258-326).  Replaces hs_join_build / hs_join_count / hs_join_fill when both key columns are INTEGER and the
 * build keys span at most `slots` <= 2^29 consecutive values from key_min (n_build < 2^31): the build rows are moved by two
 * stable range partition passes and assembled partition by partition in LDS - no global atomic, no hash table - into
 *   words[slots]        per key SLOT: 0xffffffff = no build row, a build row (the key's only one), or 0x80000000 + the
 *                       start of the key's list in rows[] (several rows);
 *   rows[n_build]       the build rows in slot order, ascending within a slot;
 *   list_count[n_build] at the start of a list of several rows: its length (other entries are not written).
 * hs_join_dense_count: ONE scattered 4-byte read per probe row (its slot word; two more only for keys with several
 * partners); counts[i] = matches of probe row i, and in aux (hs_join_dense_aux_bytes(n_probe), 16-byte aligned) the
 * first matching build row and the list start of every probe row, written sequentially.  The caller scans the counts
 * (hs_exclusive_scan_i64) and sizes the pair lists; hs_join_dense_fill writes the pairs ordered by probe row, then build
 * row - the reference's emission order - as a stream over out_start and aux: it returns to rows[] only for probe rows with
 * several partners.  A build key outside the declared range raises HS_FLAG_BAD_PROGRAM.
 * ws: hs_join_dense_ws_bytes(n_build, slots) (0 = this shape is not held). */
size_t hs_join_dense_ws_bytes(int64_t n_build, int64_t slots);
int hs_join_dense_build(void* stream, const int32_t* build_keys, int64_t n_build, int32_t key_min, int64_t slots, uint32_t* words,
                        uint32_t* rows, uint32_t* list_count, void* ws, uint32_t* flags);
size_t hs_join_dense_aux_bytes(int64_t n_probe);
int hs_join_dense_count(void* stream, const int32_t* probe_keys, int64_t n_probe, int32_t key_min, int64_t slots,
                        const uint32_t* words, const uint32_t* rows, const uint32_t* list_count, int64_t* counts, void* aux);
int hs_join_dense_fill(void* stream, int64_t n_probe, const uint32_t* rows, const void* aux, const int64_t* out_start,
                       int64_t* out_left, int64_t* out_right);
/* ---- the general inner join on ANY INTEGER keys (round 4; csrc/hs_radix.hip) -------------------------------------------
 * The same reference loop (tasks.py:201-240) for keys the dense form does not hold: a sparse or huge key range, negative
 * keys.  table = hs_join_hash_slots(n_build) 8-byte slots {key, word} (word as in the dense form), cut into windows of 512
 * slots (1024 past 19 M build rows); a key hashes to ONE window and probes linearly inside it.  Build: two stable partition passes bring the (key, row)
 * tuples into window order, one wave per window inserts / counts / scans / places them in LDS and stores the finished window
 * with coalesced stores - no global atomic, no scattered store, lists ascending without a sort.  hs_join_hash_count: one
 * scattered 8-byte read per probe row in the usual case (~1.7 slots from the start at the table's load of 0.57, nearly always
 * the same 64-byte line); counts / aux exactly as hs_join_dense_count leaves them, so the second pass IS
 * hs_join_dense_fill(rows, aux, ...).  n_build <= ~38 M rows per call (hs_join_hash_slots returns 0 beyond); a window with
 * more distinct keys than slots (a degenerate hash) is left empty and raises HS_FLAG_DICT_FULL - the caller takes
 * hs_join_build instead.  rows / list_count: n_build entries each; ws: hs_join_hash_ws_bytes(n_build). */
size_t hs_join_hash_ws_bytes(int64_t n_build);
int64_t hs_join_hash_slots(int64_t n_build);
int hs_join_hash_build(void* stream, const int32_t* build_keys, int64_t n_build, void* table, uint32_t* rows,
                       uint32_t* list_count, void* ws, uint32_t* flags);
int hs_join_hash_count(void* stream, const int32_t* probe_keys, int64_t n_probe, int64_t n_build, const void* table,
                       const uint32_t* rows, const uint32_t* list_count, int64_t* counts, void* aux);
/* Merge order of a multi-rank final aggregate (the reference reads a partition's shuffle files in block order,
 * tasks.py:117-133): STABLE sort of positions 0 .. n-1 by order[i] in [-1, n_order) (global block id; -1 = padding,
 * sorted first) with the radix tier's partition passes, least significant byte first.  out_perm[j] = position of the
 * j-th row in merge order, out_sorted[j] = its order key.  ws: hs_sort_by_order_ws_bytes(n). */
size_t hs_sort_by_order_ws_bytes(int64_t n);
int hs_sort_by_order(void* stream, const int64_t* order, int64_t n, int64_t n_order, int64_t* out_perm, int64_t* out_sorted,
                     void* ws);
/* out[i] = values[s] for bounds[s] <= i < bounds[s+1], i in [0, n) (device arrays; bounds has n_seg + 1 entries,
 * bounds[0] = 0): the global block id of every partial row of a multi-rank partial aggregate. */
int hs_expand_by_bounds(void* stream, const int64_t* bounds, const int64_t* values, int64_t n_seg, int64_t n, int64_t* out);
/* Debug aid: with HIPSPARK_RADIX_STAMPS=1 in the environment the fold kernel sums the cycles its waves spend per phase;
 * out8 = {clear tables, wait for loads, slot lookup, ranking, fold, emit, waves, 0} since the last call. */
int hs_group_radix_debug_stamps(uint64_t* out8);
/* Debug aid: with HIPSPARK_SCAN_STAMPS=1 in the environment every workgroup of the private-table scan (hs_agg_partial*)
 * leaves sixteen words: wall_clock64() (100 MHz) at [0] entry, [1] after the early-exit check, [2] after the table initialisation,
 * [3] after its last step, [4] after the table reduction, [5] after the arrival count, [6] at exit, [7] (XCC_ID << 32 | HW_ID),
 * [8..13] inside the unit combine of a last arriver (initialised, first batch's entries filed, its partials staged, folded,
 * all batches folded, rows written), [14] all waves past their last step.  Copies the
 * last launch's words of up to max_chunks workgroups to host_out; returns the number of workgroups copied (0: stamps off). */
int64_t hs_agg_debug_scan_stamps(int64_t* host_out, int64_t max_chunks);
/* mask[s] = 1 for non-empty slots (compact it with hs_compact to get the dense slot list). */
int hs_group_mask(void* stream, const int64_t* slot_start, int64_t table_cap, uint8_t* mask);
/* One lane per group folds val_cols[a][position] over the group's positions front to back - the reference's
 * own order (fill_aggregators, tasks.py:295-310), so fp64 sums are bit-identical to Python's.  Outputs:
 * out_rep_row[g] = row id of the group's first row, out_acc[a * n_groups_max + g] = 64-bit cell (quantised
 * like a shuffle-file write when quantise != 0). */
int hs_group_fold(void* stream, const hs_col* val_cols, const hs_agg_spec* spec, const int64_t* slot_list,
                  int64_t n_groups_max, const int64_t* n_groups_dev, const int64_t* slot_start,
                  const int64_t* positions, const int64_t* sel, int64_t row0, int32_t quantise, int64_t* out_rep_row,
                  uint64_t* out_acc, uint32_t* flags);

/* Probe pass 1: match count per right row -> counts[n_right]. */
int hs_join_count(void* stream, const hs_col* left_key, const hs_col* right_key, int64_t n_right,
                  int64_t table_cap, const uint64_t* table_keys, const int64_t* table_reps,
                  const int64_t* slot_start, int64_t* counts);
/* Probe pass 2 (after an exclusive scan of counts into out_start[n_right+1]): emit (left,right) row
 * pairs in the reference's order: right rows ascending, for each its left matches ascending
 * (tasks.py:224-240). */
int hs_join_fill(void* stream, const hs_col* left_key, const hs_col* right_key, int64_t n_right,
                 int64_t table_cap, const uint64_t* table_keys, const int64_t* table_reps,
                 const int64_t* slot_start, const int64_t* rows, const int64_t* out_start, int64_t* out_left,
                 int64_t* out_right);
/* ---- primary-key / foreign-key join on INTEGER keys (round 2; csrc/hs_join.hip) --------------------------------
 * The build side's keys are expected to be unique (else HS_FLAG_JOIN_DUP is raised and the caller falls back to
 * hs_join_build / count / fill).  table: `slots` 32-bit words, one build row per slot.  direct != 0: slot =
 * key - key_min (slots = the key range; chosen by the caller when the range is dense - TPC-H order keys use 8 of
 * every 32 values - plain stores, then a streaming count of the occupied slots: fewer than n_build = two rows met
 * in one slot; no atomics on the table; the table buffer must hold slots + 4 words, the count lives behind it);
 * else slots = a power of two >= 2 * n_build, multiplicative hashing + linear probing, one 32-bit CAS per build
 * row.  minmax: device int32[2]; values 16-byte aligned, read up to 3 elements past n. */
int hs_minmax_i32(void* stream, const int32_t* values, int64_t n, int32_t* minmax);
int hs_join_build_unique(void* stream, const int32_t* build_keys, int64_t n_build, int32_t key_min, int64_t slots,
                         int32_t direct, uint32_t* table, uint32_t* flags);
/* Probe rows stay where they are.  Per probe row i: out_build_row[i] (optional) = the matching build row (0 without a
 * match), out_unit[i] = python_hash(key) % n_parts (the row's shuffle partition, tasks.py:362 - the unit id for
 * hs_agg_shared_units) or 0xff without a match (inner join: the row is dropped), out_payload[i] (optional) =
 * build_payload[build row], a 1-byte column of the build side gathered on the way (dictionary codes of a GROUP BY
 * key).  probe_keys and the outputs must be 16- / 4-byte aligned; probe_keys may be read up to 3 elements past
 * n_probe (buffers carry slack). */
int hs_join_probe_unique(void* stream, const int32_t* probe_keys, int64_t n_probe, const int32_t* build_keys,
                         int32_t key_min, int64_t slots, int32_t direct, const uint32_t* table, int32_t n_parts,
                         int64_t* out_build_row, uint8_t* out_unit, const uint8_t* build_payload, uint8_t* out_payload);

/* ---- the same join with the PAYLOAD IN THE TABLE and the probe inside the aggregate (round 3) ----------------------
 * Reference loops replaced: BroadcastHashJoinTask.generate_chunks tasks.py:201-240 (build the left side's hash map,
 * probe with every right row, emit joined rows) feeding AggregateTask tasks.py:284-289 per JoinJob (plan.py:99-109).
 * Table: ONE BYTE per slot, slot = key - key_min, value = the build row's 1-byte payload (the dictionary code of the
 * one build-side column the aggregate reads; 0 when it reads none) or 0xff = no such key.  sf=10 orders: 60 MB instead
 * of 240 MB of row numbers + a second dependent read of the payload column; it stays in the Infinity Cache.
 *
 * hs_join8_build: the build rows arrive in any order, so byte stores into the table would each cost a 32-byte HBM
 * sector (round 2 measured 7.8x write amplification).  Instead: (1) per workgroup a histogram of the rows' WINDOWS
 * (HS_JOIN8_WINDOW consecutive slots), (2) a scan, (3) the rows are partitioned by window as 4-byte (offset, payload)
 * tuples, (4) one workgroup per window assembles its slice of the table in LDS and writes it out with 16-byte
 * stores - the table is written exactly once, coalesced, and never read back: two tuples meeting in one byte
 * (= a duplicate key -> HS_FLAG_JOIN_DUP) show up as fewer occupied bytes than tuples, counted while writing.
 * Build rows may be the concatenation of equally long SEGMENTS (the all-gathered shares of N ranks): seg_len rows
 * each (a multiple of 4), of which the first seg_counts[s] (device, optional) are valid; n_build = segments * seg_len.
 * payload NULL: every present key gets byte 0.  payload bytes must be < 0xff.  A key outside [key_min, key_min +
 * slots) raises HS_FLAG_BAD_PROGRAM and is left out.  table: hs_join8_table_bytes(slots) bytes, 16-byte aligned;
 * ws: hs_join8_ws_bytes(n_build, slots).  slots <= 2^30. */
#define HS_JOIN8_WINDOW 65536
size_t hs_join8_table_bytes(int64_t slots);
size_t hs_join8_ws_bytes(int64_t n_build, int64_t slots);
int hs_join8_build(void* stream, const int32_t* build_keys, const uint8_t* payload, int64_t n_build, int64_t seg_len,
                   const int64_t* seg_counts, int32_t key_min, int64_t slots, uint8_t* table, void* ws, uint32_t* flags);
/* ---- N ranks: the build side sharded by the probe side's key stripes (round 4).  Reference: both join inputs are
 * shuffled on the key (plan.py:186-189), one JoinJob per partition (plan.py:99-109, tasks.py:201-240).  Here probe
 * rows stay on the rank that owns their file block; a probe table clustered on the key gives every block a key stripe
 * [min, max] (hs_minmax_i32_units: minmax[2u], minmax[2u + 1] per unit; a unit without rows: INT32_MAX, INT32_MIN), and
 * a build row travels to the owner of every stripe that contains its key: none, one, or the two neighbours of a block
 * boundary.  stripe_min / stripe_max / stripe_owner [n_stripes <= 4096] (device) list the non-empty blocks of ALL ranks in
 * block order, min and max non-decreasing (the caller checks; otherwise it keeps the all-gathered build).
 * hs_join8_route_count -> dest_start[world + 1] (device, caller-owned): where each destination's share of the routed
 * (row, destination) pairs begins.  hs_join8_route (same ws and dest_start, after the count): out_keys / out_codes grouped by
 * destination, ready for all_to_all_single with those split sizes; expect_start (optional, device): split sizes agreed in
 * an earlier run - if the data routes differently now nothing is written and HS_FLAG_ROUTE_STALE is raised.
 * hs_join8_build_windows: hs_join8_build over the received rows (no segments), assembling and writing ONLY the windows
 * with window_mask[w] != 0 (device, one byte per HS_JOIN8_WINDOW slots) - those the rank's own stripes reach. */
int hs_minmax_i32_units(void* stream, const int32_t* values, const int64_t* unit_rows_dev, int64_t n_units, int32_t* minmax);
size_t hs_join8_route_ws_bytes(int64_t n_build, int32_t world);
int hs_join8_route_count(void* stream, const int32_t* build_keys, int64_t n_build, const int32_t* stripe_min,
                         const int32_t* stripe_max, const int32_t* stripe_owner, int32_t n_stripes, int32_t world, void* ws,
                         uint32_t* dest_start);
int hs_join8_route(void* stream, const int32_t* build_keys, const uint8_t* payload, int64_t n_build, const int32_t* stripe_min,
                   const int32_t* stripe_max, const int32_t* stripe_owner, int32_t n_stripes, int32_t world, void* ws,
                   const uint32_t* dest_start, const uint32_t* expect_start, int32_t* out_keys, uint8_t* out_codes, int64_t out_cap,
                   uint32_t* flags);
int hs_join8_build_windows(void* stream, const int32_t* build_keys, const uint8_t* payload, int64_t n_build, int32_t key_min,
                           int64_t slots, const uint8_t* window_mask, uint8_t* table, void* ws, uint32_t* flags);
/* out[i] = lut[codes[i]] (re-coding a dictionary-coded column after the ranks agreed on one dictionary). */
int hs_remap_u8(void* stream, const uint8_t* codes, int64_t n, const uint8_t* lut256, uint8_t* out);

/* ---- dictionary-coded STRING columns (round 2; csrc/hs_join.hip) ------------------------------------------------
 * A STRING column with at most 256 distinct values is re-coded at table open as one byte per row + its dictionary
 * (reference strings stay what they are on disk: io.py:97-109).  LIKE / comparisons with literals then become a
 * bit test on the code (HS_OP_DICTBIT), string concatenation a mixed-radix combination of codes, GROUP BY a
 * 1-byte key.  hs_dict_build inserts every row's string into an open-addressing set (cap slots, power of two >=
 * 512; slot_words[cap], slot_reps[cap] = a row holding the slot's string or -1; *count = occupied slots;
 * HS_FLAG_DICT_FULL in *flags once the table is half full - the column is then left as it is).  Under contention a
 * string may occupy more than one slot: the caller reads the slots' strings back through slot_reps, names codes BY
 * STRING (duplicate slots share a code; more than 256 distinct strings = no encoding), uploads slot_code[cap] and
 * calls hs_dict_assign, which looks every row's string up again (any of its slots gives the same code). */
int hs_dict_build(void* stream, const hs_col* col, int64_t nrows, int32_t cap, uint64_t* slot_words, int64_t* slot_reps,
                  int32_t* count, uint32_t* flags);
int hs_dict_assign(void* stream, const hs_col* col, int64_t nrows, int32_t cap, uint64_t* slot_words, int64_t* slot_reps,
                   const uint8_t* slot_code, uint8_t* out_codes, uint32_t* flags);
/* out[i] = sum_k codes[k][i] * strides[k] (the code of a concatenation in the product dictionary); 1..4 parts,
 * 16-byte aligned buffers that may be read up to 15 bytes past nrows. */
int hs_dict_combine(void* stream, int32_t n_parts, const uint8_t* const* codes, const int32_t* strides, int64_t nrows,
                    uint8_t* out_codes);

/* Exclusive scan of int64 counts -> start[n+1] (start[n] = total).  ws: hs_scan_ws_bytes(n). */
int hs_exclusive_scan_i64(void* stream, const int64_t* counts, int64_t n, int64_t* start, void* ws);

/* =================================================================================================
 * A6/K12  Quantisation at a file write (reference io.py:87-94)
 * ===============================================================================================*/

/* f64 -> f32 (RNE; finite overflow sets HS_FLAG_FLT_OVERFLOW) or i64 -> i32 (range check sets
 * HS_FLAG_INT_OVERFLOW).  src_kind in {HS_F64, HS_I64}. */
int hs_quantise(void* stream, const void* src, int32_t src_kind, int64_t n, const int64_t* n_dev, void* dst,
                uint32_t* flags);
/* The same for up to 16 columns of one batch in a single launch. */
int hs_quantise_many(void* stream, int32_t n_cols, void* const* srcs, const int32_t* src_kinds, int64_t n,
                     const int64_t* n_dev, void* const* dsts, uint32_t* flags);

/* =================================================================================================
 * The short tail: partial rows written straight into the exchange slab, and ONE launch from the
 * (all-gathered) slabs to the query's result.  For GROUP BY queries whose partial rows fit the LDS tiers
 * the work after the scan is latency-bound (a few hundred rows); this pair replaces
 * hs_agg_pack + key gather + [all-gather] + hs_slab_unpack + hs_agg_merge + key gather + hs_eval +
 * hs_quantise_many + concatenation by two launches around the collective.  Same arithmetic, same order.
 *
 * Slab of one rank (hs_slab_desc; host mirror: minispark_amd/distributed.py SlabLayout):
 *   [flags u32][pad u32][row count i64] | order key i64 x M | key column M x key_bytes | accumulator
 *   columns M x 4 bytes (HS_F32 / HS_I32 = what the reference's shuffle file holds, io.py:87-94).
 * Unit u owns rows [u * group_cap, (u + 1) * group_cap): its groups dense from the first row, order key =
 * the unit's global id; unused rows carry order key -1.
 * ===============================================================================================*/
typedef struct hs_slab_desc {
    int64_t slab_rows;           /* M */
    int64_t stride;              /* bytes from one rank's slab to the next in the gathered buffer */
    int64_t order_off, key_off;  /* byte offsets inside a slab */
    int64_t acc_off[HS_MAX_ACC];
    int32_t key_kind;            /* HS_I32 / HS_F32 / HS_I64 / HS_F64 / HS_U8, or HS_STR of fixed length key_len */
    int32_t key_len;             /* HS_STR: 1, 2 or 4 bytes (packs into the 64-bit key word); else ignored */
    int32_t n_acc, pad;
    int32_t acc_kind[HS_MAX_ACC]; /* HS_F32 / HS_I32 */
} hs_slab_desc;

/* hs_agg_partial with the unit combine emitting into `slab` (this rank's slab, device memory, order keys
 * pre-set to -1, header zero).  unit_ids (device, optional) = global id of every local unit.  Status bits of
 * the scan and the combine are OR-ed into the slab header as well as into *flags, so they reach every rank. */
int hs_agg_partial_slab(void* stream, const hs_col* cols, int32_t n_cols, int32_t key_col, const hs_program* prog,
                        const hs_agg_spec* spec, const hs_chunk* chunks, const int64_t* unit_chunk0, int64_t n_units,
                        const hs_agg_geom* geom, const int64_t* unit_ids, uint8_t* slab, const hs_slab_desc* desc,
                        void* ws, uint32_t* flags, void* ev_begin, void* ev_end);

/* One result column of hs_agg_finish. */
typedef struct hs_finish_out {
    int32_t src;     /* 0 = the group key, 1 = merged aggregate `index` (fold number), 2 = output `index` of the program */
    int32_t index;
    int32_t kind;    /* stored kind: HS_F32 (from an f64 cell, overflow flagged), HS_I32 (from i64, range flagged),
                        HS_I64 (as is); src 0: ignored - the key element is copied */
    int32_t pad;
    int64_t offset;  /* byte offset of the column inside `result` (cap elements) */
} hs_finish_out;
#define HS_FINISH_MAX_OUT 24
typedef struct hs_finish_spec {
    int32_t n_fold;                   /* merged aggregates: fold j = fold_op[j] over slab accumulator fold_src[j] */
    int32_t fold_src[HS_MAX_ACC];
    int32_t fold_op[HS_MAX_ACC];      /* HS_AGG_SUM / MIN / MAX (reference tasks.py:290-292) */
    int32_t n_out;
    hs_finish_out outs[HS_FINISH_MAX_OUT];
    int32_t prog_src[HS_MAX_COLS];    /* program column slot -> -1 = the group key, j = merged aggregate j */
    int32_t prog_out[HS_MAX_OUTS];    /* program output -> index into outs[] */
} hs_finish_spec;
size_t hs_agg_finish_scratch_bytes(int32_t cap, int32_t n_fold);
/* Final merge of the partial rows of `world` slabs (gathered: world slabs desc->stride bytes apart; world 1 =
 * the rank's own slab) in ascending (order key, row) order exactly like hs_agg_merge, then the projection
 * `prog` over the merged rows (NULL / n_ins 0: none), quantisation to the stored kinds and the result image:
 *   result := [flags u32][done u32 = 1, written last][number of groups i64] + the columns at outs[].offset.
 * flags = *flags | every slab's header flags | this launch's own bits; *flags and *own_slab_flags (optional)
 * are reset to 0 for the next run.  n_order: order keys lie in [0, n_order).  HS_E_LIMIT (2) when the rows do
 * not fit the LDS tier. */
int hs_agg_finish(void* stream, const uint8_t* gathered, int32_t world, const hs_slab_desc* desc,
                  const hs_finish_spec* fin, const hs_program* prog, int64_t n_order, int32_t cap, uint8_t* result,
                  void* scratch, uint32_t* flags, uint32_t* own_slab_flags);
/* `result` may live in pinned host memory mapped into the device (zero-copy hand-over, no device->host copy
 * in the query): hs_agg_finish stores the image with system-scope visibility and then sets the header's second
 * word (the pad) to 1, which the host may poll after clearing it.  This returns the device address of such a
 * host allocation, or an error if the memory is not device-accessible. */
int hs_host_device_pointer(void* host_ptr, void** device_ptr);

/* =================================================================================================
 * Multi-GPU: un-interleave the all-gathered exchange slabs of the partial-aggregate shuffle (the slab
 * layout is minispark_amd/distributed.py SlabLayout: [flags u32][pad][row count i64][order key i64 x M]
 * [column 0: M x row_bytes] ...).  gathered = [world][slab_bytes].  Outputs: flags_out[world];
 * order_out[world * M] (-1 on rows at or beyond a rank's count); every column contiguous over world * M rows
 * (rank-major) in col_dsts[c].  Replaces the reference's shuffle-file read-back (tasks.py:144-150).
 * ===============================================================================================*/
int hs_slab_unpack(void* stream, const uint8_t* gathered, int32_t world, int64_t slab_bytes, int64_t slab_rows,
                   int64_t order_offset, int32_t n_cols, const int64_t* col_offsets, const int32_t* col_row_bytes,
                   void* const* col_dsts, int32_t* flags_out, int64_t* order_out);

/* =================================================================================================
 * Multi-GPU: pack / unpack of the generic row exchange (round 3; csrc/hs_exchange.hip).  A rank's rows for all
 * peers travel in ONE byte buffer (all_to_all_single): destination-major, inside a destination's share one slice per
 * column piece (fixed-width values; length bytes and payload bytes of a STRING column).  Replaces the reference's
 * per-partition shuffle files, written row by row (tasks.py:347-375) and concatenated by the reader
 * (tasks.py:144-150).  Both directions are one launch over a device-resident list of segments:
 * dst[0 .. bytes) = src[0 .. bytes) for every segment; max_bytes = the largest segment (sizes the launch).
 * ===============================================================================================*/
typedef struct hs_segment {
    const void* src;
    void* dst;
    int64_t bytes;
} hs_segment;
int hs_copy_segments(void* stream, const hs_segment* segments_dev, int32_t n_segments, int64_t max_bytes);

/* Peer-to-peer exchange of the short tail's slabs (prototype; csrc/hs_exchange.hip): instead of an all-gather every rank
 * stores its slab straight into a buffer of every peer (mapped through hipIpc handles by the host) and raises a flag there;
 * the receiver waits for `world` flags on the device and lays the slots out as the gathered slabs hs_agg_finish reads.
 * A rank's buffer: hs_slab_p2p_bytes(world, slot_bytes) bytes, zeroed once; slot_bytes a multiple of 16 >= the slab.
 * peers_dev: device array of `world` pointers - every rank's buffer as mapped in this process (own buffer at [rank]);
 * epochs_dev: two device uint64, zeroed once ([0] pushes, [1] waits of this rank).  The wait gives up after timeout_ms
 * with HS_FLAG_PEER_TIMEOUT in *flags.  Replaces the reference's shuffle-file hand-over (tasks.py:347-375, 144-150) for
 * the partial rows of a GROUP BY. */
size_t hs_slab_p2p_bytes(int32_t world, int64_t slot_bytes);
int hs_slab_push(void* stream, const void* slab, int64_t slab_bytes, void* const* peers_dev, int32_t world, int32_t rank,
                 int64_t slot_bytes, uint64_t* epochs_dev);
int hs_slab_wait(void* stream, void* own_buf, int32_t world, int64_t slot_bytes, int64_t slab_bytes, uint64_t* epochs_dev,
                 void* gathered, int64_t out_stride, uint32_t* flags, int64_t timeout_ms);

/* =================================================================================================
 * Run-time specialisation (reference: codegen.py:230-247 compiles every query with `zig build`).
 * hs_agg_partial translates its bytecode to straight-line code inside the hand-written kernel skeleton,
 * compiles it with hiprtc for the running GPU and caches it per program; on any failure it launches the
 * ahead-of-time interpreter kernel instead (still on the GPU).  Env HIPSPARK_JIT=0 disables it.
 * ===============================================================================================*/
void hs_jit_set_enabled(int enabled);
int hs_jit_get_enabled(void);
/* counters[0] = programs compiled, [1] = launches of compiled programs, [2] = compile failures */
void hs_jit_stats(int32_t* counters);
/* Compiled code objects are kept on disk between processes: $HIPSPARK_JIT_CACHE (default $TMPDIR/hipspark-jit-<uid>,
 * "0" = off), one file per program named by a hash of everything the compiler sees; -> programs of this process that
 * were loaded from there instead of being compiled. */
int hs_jit_disk_hits(void);
/* Seconds this process has spent producing code objects (hiprtc compiles + cache reads) - the reference's counterpart is
 * the wall time of its per-query `zig build` (codegen.py:239-245). */
double hs_jit_compile_seconds(void);
const char* hs_jit_last_log(void);
/* Translate + compile only (needs no GPU): proves the generated source builds for `arch`
 * (NULL = "gfx950"); optionally returns the generated source text. */
int hs_jit_compile_check(const hs_col* cols, int32_t n_cols, int32_t key_col, const hs_program* prog,
                         const hs_agg_spec* spec, const char* arch, int64_t* code_bytes, char* src_out,
                         int64_t src_cap);
/* The same for the shared-dictionary tier's kernel (hs_agg_shared / hs_agg_shared_units; unit_col = -1: row-range
 * units). */
int hs_jit_compile_check_shared(const hs_col* cols, int32_t n_cols, int32_t key_col, int32_t unit_col,
                                const hs_program* prog, const hs_agg_spec* spec, const char* arch, int64_t* code_bytes,
                                char* src_out, int64_t src_cap);
/* The same for an expression program of hs_eval (the compiled form evaluates four rows per lane with 16-byte
 * loads and stores; hs_eval uses it when sel == NULL and the numeric buffers are 16-byte aligned, and then reads
 * up to 3 rows past nrows of every column it loads - buffers carry that slack). */
int hs_jit_compile_check_eval(const hs_col* cols, int32_t n_cols, const hs_program* prog, const int32_t* out_kinds,
                              int32_t n_outs, const char* arch, int64_t* code_bytes, char* src_out, int64_t src_cap);

/* =================================================================================================
 * Stage-level ABI (csrc/hs_engine.hip): a GROUP BY query over one BlockFile table, end to end, without Python.
 * Replaces the reference's process boundary - jobs shipped to a worker over stdin, src/mini_spark/execution.py:182-219,
 * jobs.py:45-79, zig-src/src/job.zig:3-57 - for the hot path: the host (Python engine, cgo / JNI / FFI binding:
 * INTEGRATION.md section 4) lowers the query once into a PLAN BLOB and then calls
 *
 *   hs_engine_create -> hs_table_open -> hs_stage_prepare -> hs_stage_run (any number of times)
 *                    -> hs_result_columns / hs_result_write_blockfile -> hs_stage_destroy / hs_table_close / hs_engine_destroy
 *
 * The library owns everything in between: the native BlockFile reader (header / footer / column spans, column pruning,
 * parallel pread into pinned staging, async host-to-device copies; reference block_file.zig:225-306), chunk geometry,
 * the exchange slab and result-image layouts, workspaces, the capacity retry on a dictionary overflow, the
 * steady-state replay of a run's launches, and the zero-copy result hand-over.  Queries that do not fit the on-chip
 * tiers of this path (variable-length GROUP BY keys, more partial rows than the on-chip final merge holds, > 8 numeric
 * columns) return HS_E_LIMIT.  Up to 16 groups per block: per-lane tables + the one-launch finish, replayed; more: the
 * shared-dictionary scan + hs_agg_pack / hs_agg_merge / hs_eval / hs_quantise issued by hs_stage_run itself (one GPU).
 * ===============================================================================================*/
typedef struct hs_engine hs_engine;
typedef struct hs_table hs_table;
typedef struct hs_stage hs_stage;

#define HS_STAGE_PLAN_VERSION 2
/* The plan of [scan -> WHERE -> partial aggregate per file block] + [final merge -> projection -> result write]
 * (reference plan.py:182-204), as minispark_amd/stage.py lowers it from the reference's task objects. */
typedef struct hs_stage_plan {
    int32_t version;               /* HS_STAGE_PLAN_VERSION */
    int32_t n_cols;                /* column slots of `prog` */
    int32_t col_ids[HS_MAX_COLS];  /* table column behind every slot (only these are read from the file) */
    int32_t key_slot;              /* slot of the GROUP BY column */
    int32_t group_cap, merge_cap;  /* starting dictionary capacities (powers of two; 0 = 4 / 16); grown on overflow */
    hs_program prog;               /* [filter ... FILTER]* KEY [argument ... AGG acc]* */
    hs_agg_spec spec;              /* the accumulators of `prog` */
    hs_finish_spec fin;            /* folds of the final merge + the result columns (offsets are filled in here) */
    hs_program fin_prog;           /* projection after the merge (AVG = sum / count ...); n_ins 0 = none */
    int32_t out_types[HS_FINISH_MAX_OUT];    /* BlockFile type code of every result column (0 INTEGER 1 STRING 2 FLOAT 3 TIMESTAMP) */
    char out_names[HS_FINISH_MAX_OUT][64];   /* ... and its name (hs_result_write_blockfile) */
    /* version 2: a COMPUTED GROUP BY key (SELECT (l_orderkey % 331 - 100) AS bucket ... GROUP BY bucket: a ProjectTask
     * in front of the aggregate, reference tasks.py:32-35).  col_ids[key_slot] = -1; every run evaluates key_prog (one
     * HS_OP_OUT 0, INTEGER-valued; the lowering only hands over expressions whose value provably fits the stored
     * INTEGER and cannot raise, so rows the WHERE drops may be evaluated too) over the table columns kcol_ids into a
     * 4-byte key column that lives next to the table's.  Projected columns used elsewhere are inlined into `prog`. */
    int32_t key_computed, n_kcols;
    int32_t kcol_ids[HS_MAX_COLS];
    hs_program key_prog;
} hs_stage_plan;

typedef struct hs_result_col {
    int32_t kind;        /* HS_F32 / HS_I32 / HS_I64, or HS_STR of fixed byte width `width` (the GROUP BY key) */
    int32_t width;       /* bytes per row */
    const void* data;    /* HOST memory (the stage's result image): n_rows x width bytes, valid until the next run */
    int64_t n_rows;
} hs_result_col;

int hs_engine_create(int32_t device, hs_engine** out);
void hs_engine_destroy(hs_engine* engine);
/* Wall time and bytes of the last hs_table_load's read + host-to-device pipeline on this engine (the ingest bench's
 * figure: page cache -> pinned staging -> HBM, against the PCIe link's peak). */
int hs_engine_load_stats(const hs_engine* engine, double* seconds, int64_t* bytes);
/* The reader's pipeline with caller-owned destinations: n_spans byte ranges of the file at `path` -> device addresses
 * (reader threads pread chunks into the engine's pinned staging pool while earlier chunks' H2D copies are in flight).
 * Returns when every byte has arrived.  Replaces the reference's per-block decode loop (io.py:112-163) for a host that
 * keeps its own device buffers. */
typedef struct hs_span {
    int64_t file_offset, bytes;
    void* dst; /* device address */
} hs_span;
int hs_read_spans(hs_engine* engine, const char* path, const hs_span* spans, int32_t n_spans);
/* Parses header, footer and the column spans of the blocks this rank owns (block b -> rank b % world). */
int hs_table_open(hs_engine* engine, const char* path, int32_t rank, int32_t world, hs_table** out);
void hs_table_close(hs_table* table);
int hs_table_info(const hs_table* table, int32_t* n_cols, int64_t* n_rows, int32_t* n_blocks, int32_t* total_blocks);
int hs_table_schema(const hs_table* table, int32_t col, int32_t* type, char* name, int32_t name_cap);
/* Reads only the listed columns' byte spans into HBM (idempotent per column). */
int hs_table_load(hs_engine* engine, hs_table* table, const int32_t* col_ids, int32_t n);
int hs_table_column(const hs_table* table, int32_t col, hs_col* out, int64_t* n_rows);
/* Caller-owned device columns as a table (types: BlockFile type codes; cols[c].data NULL = column absent). */
int hs_table_attach(hs_engine* engine, int32_t n_cols, const hs_col* cols, const int32_t* types, const int64_t* block_rows,
                    int32_t n_blocks, hs_table** out);
/* plan_bytes must be sizeof(hs_stage_plan) (a binding's mirror is checked that way); loads the plan's columns. */
int hs_stage_prepare(hs_engine* engine, hs_table* table, const hs_stage_plan* plan, size_t plan_bytes, int32_t world,
                     hs_stage** out);
void hs_stage_destroy(hs_stage* stage);
/* world 1: launches, waits for the result image, retries with larger dictionaries on HS_FLAG_DICT_FULL (per-unit
 * tables) / HS_FLAG_MERGE_FULL (final merge).  *flags_out:
 * the remaining HS_FLAG_* bits (division by zero, overflow at a file write ...) for the host to raise. */
int hs_stage_run(hs_stage* stage, void* stream, uint32_t* flags_out, int64_t* n_rows_out);
/* world > 1: scan into this rank's slab | the caller all-gathers the slabs (RCCL) | finish over the gathered slabs. */
int hs_stage_launch_partial(hs_stage* stage, void* stream);
void* hs_stage_slab(hs_stage* stage, int64_t* bytes);
int hs_stage_launch_finish(hs_stage* stage, void* stream, const void* gathered, int32_t world);
int hs_stage_wait(hs_stage* stage, void* stream, uint32_t* flags_out, int64_t* n_rows_out);
int hs_stage_grow(hs_stage* stage);
/* stats[6]: runs, replayed runs, capacity growths, group_cap, merge_cap, scan chunks */
int hs_stage_stats(const hs_stage* stage, int64_t* stats);
int hs_result_columns(const hs_stage* stage, hs_result_col* out, int32_t cap, int32_t* n);
int hs_result_write_blockfile(const hs_stage* stage, const char* path);

/* ---- the JOIN stage behind the same boundary (round 3) --------------------------------------------------------------
 * The reference's JoinJob per shuffle partition (jobs.py:45-79, plan.py:99-109: tasks.py:201-240 build + probe feeding
 * the partial aggregate tasks.py:284-289) and the final stage after it, for the primary-key / foreign-key case of
 * DESIGN.md 4.6, end to end without Python: both tables through the BlockFile reader, dictionary coding of the ONE
 * build-side column the aggregate reads, key range, byte table (hs_join8_build), probe inside the aggregate scan
 * (hs_agg_shared_join8), raw unit tables -> exchange slab -> finish launch -> result image -> result BlockFile; capacity
 * retry and steady-state replay like hs_stage_run.  HS_E_LIMIT for anything else (non-INTEGER or duplicate build keys, a
 * sparse key range, a second build-side column, a GROUP BY key wider than 4 bytes): those belong to the per-operator ABI. */
typedef struct hs_join_stage hs_join_stage;
#define HS_JOIN_STAGE_PLAN_VERSION 1
typedef struct hs_join_stage_plan {
    int32_t version;               /* HS_JOIN_STAGE_PLAN_VERSION */
    int32_t build_key_col;         /* build table: INTEGER column holding every key once */
    int32_t build_payload_col;     /* build table: the STRING column the aggregate reads (<= 255 distinct values), or -1 */
    int32_t probe_key_col;         /* probe table: INTEGER column */
    int32_t n_parts;               /* shuffle partitions = JoinJobs (the reference's SHUFFLE_PARTITIONS, 10) */
    int32_t n_cols;                /* column slots of `prog` (the unit column is added behind them) */
    int32_t col_ids[HS_MAX_COLS];  /* slot -> probe table column, or -1 = the build-side column (its dictionary code) */
    int32_t key_slot;              /* slot of the GROUP BY column */
    int32_t group_cap, merge_cap;  /* starting capacities per JoinJob / of the final merge (0 = 4 / 16); grown on overflow */
    hs_program prog;               /* [filter ... FILTER]* KEY [argument ... AGG acc]* over the slots */
    hs_agg_spec spec;
    hs_finish_spec fin;            /* as in hs_stage_plan */
    hs_program fin_prog;
    int32_t out_types[HS_FINISH_MAX_OUT];
    char out_names[HS_FINISH_MAX_OUT][64];
} hs_join_stage_plan;
int hs_join_stage_prepare(hs_engine* engine, hs_table* build, hs_table* probe, const hs_join_stage_plan* plan, size_t plan_bytes,
                          hs_join_stage** out);
int hs_join_stage_run(hs_join_stage* stage, void* stream, uint32_t* flags_out, int64_t* n_rows_out);
/* stats[8]: runs, replays, grows, group_cap, merge_cap, dictionary entries, table slots, slots per unit table */
int hs_join_stage_stats(const hs_join_stage* stage, int64_t* stats);
int hs_join_result_write_blockfile(const hs_join_stage* stage, const char* path);
void hs_join_stage_destroy(hs_join_stage* stage);

/* ---- the SELECT / WHERE stage behind the same boundary (round 3) -------------------------------------------------------
 * A ScanJob whose rows go to the result file (jobs.py:45-60; FilterTask tasks.py:167-177, ProjectTask tasks.py:32-35,
 * WriteToLocalFileTask tasks.py:391-410): table -> [WHERE] -> selected columns.  An output is either a table column passed
 * through as stored, or output k of the `project` program (a numeric expression, evaluated in f64 / i64 over the surviving
 * rows and rounded to the file's FLOAT / INTEGER with the reference's overflow errors).  The result is held on the host
 * after a run; hs_select_result_write_blockfile writes it as blocks of rows_per_block rows (the reference: 2 097 152). */
typedef struct hs_select_stage hs_select_stage;
#define HS_SELECT_STAGE_PLAN_VERSION 1
typedef struct hs_select_stage_plan {
    int32_t version;                       /* HS_SELECT_STAGE_PLAN_VERSION */
    int32_t n_cols;                        /* column slots of `filter` */
    int32_t col_ids[HS_MAX_COLS];          /* slot -> table column */
    hs_program filter;                     /* one HS_OP_OUT 0 = the row survives (HS_U8 mask); n_ins 0 = no WHERE */
    int32_t n_pcols;                       /* column slots of `project` */
    int32_t pcol_ids[HS_MAX_COLS];
    hs_program project;                    /* HS_OP_OUT k = computed column k; n_ins 0 = none */
    int32_t project_kinds[HS_MAX_OUTS];    /* HS_F64 / HS_I64 per computed column */
    int32_t n_out;
    int32_t out_src[HS_FINISH_MAX_OUT];    /* >= 0: table column passed through; -1 - k: computed column k */
    int32_t out_types[HS_FINISH_MAX_OUT];  /* BlockFile type code of every result column */
    char out_names[HS_FINISH_MAX_OUT][64];
} hs_select_stage_plan;
int hs_select_stage_prepare(hs_engine* engine, hs_table* table, const hs_select_stage_plan* plan, size_t plan_bytes,
                            hs_select_stage** out);
int hs_select_stage_run(hs_select_stage* stage, void* stream, uint32_t* flags_out, int64_t* n_rows_out);
int hs_select_result_write_blockfile(const hs_select_stage* stage, const char* path, int64_t rows_per_block);
void hs_select_stage_destroy(hs_select_stage* stage);

/* =================================================================================================
 * Launch capture: the native replay of a recorded query (reference: the Zig worker re-runs its compiled plan per job,
 * zig-src/src/job.zig:3-57; here a whole query's device work is re-issued by one call).  Between hs_capture_begin()
 * and hs_capture_end() every kernel launch / event record / memset the library makes ON THIS THREAD is executed as
 * usual and also appended to a list with a private copy of its arguments; hs_capture_replay() issues that list on
 * `stream` - no argument marshalling, validation or JIT look-ups - and is valid while the buffers the captured calls
 * were given stay alive and in place.  *n_ops (optional) = launches captured.
 * ===============================================================================================*/
/* Per-launch GPU slices for the query trace (reference utils.py:85-135 wraps stages and jobs in perfetto slices and
 * merges the workers' traces, :69-79): between hs_trace_begin(stream) and hs_trace_end(stream, ...) every launch of
 * the library on THIS thread - first runs and captured replays alike - is bracketed by an event pair on the launch
 * stream.  hs_trace_end waits for the stream and returns the launches in order: kernel name, start and duration in
 * microseconds relative to the moment of hs_trace_begin on the stream. */
typedef struct hs_trace_slice {
    char name[96];
    double start_us, dur_us;
} hs_trace_slice;
int hs_trace_begin(void* stream);
int hs_trace_end(void* stream, hs_trace_slice* out, int32_t cap, int32_t* n);

int hs_capture_begin(void);
int hs_capture_end(void** handle, int32_t* n_ops);
int hs_capture_replay(void* handle, void* stream);
void hs_capture_free(void* handle);

/* =================================================================================================
 * Synthetic TPC-H-shaped data (bench / tests only).  Counter-based: the value of
 * row i depends only on (seed, column, i), so any row range can be generated independently.
 * ===============================================================================================*/
int hs_gen_lineitem(void* stream, uint64_t seed, int64_t row0, int64_t nrows, float* quantity, float* extendedprice,
                    float* discount, float* tax, int64_t* shipdate, uint8_t* returnflag, uint8_t* flag_lens,
                    int32_t* orderkey, uint8_t* shipmode_code);

/* Synthetic orders table of BASELINE config 4 (CPU twin: oracle/q45_oracle.c q4_gen_orders): o_orderkey = key(perm(row))
 * for an affine bijection perm of [0, n_total), priority_code in [0, 5).  n_total < 2^31. */
int hs_gen_orders(void* stream, uint64_t seed, int64_t row0, int64_t nrows, int64_t n_total, int32_t* orderkey,
                  uint8_t* priority_code);

#ifdef __cplusplus
}
#endif
#endif /* HIPSPARK_H */
