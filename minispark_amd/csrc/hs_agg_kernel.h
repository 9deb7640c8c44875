// hs_agg_kernel.h - body of the fused scan + WHERE + aggregate-argument + partial-aggregate kernel,
// templated on a `Prog` that says how one row quad is loaded and evaluated:
//
//   * InterpProg<HASHED, D>  (compiled ahead of time in hs_agg.hip): runs the stack bytecode;
//   * a generated JitProg    (compiled at run time by hs_jit.cpp through hiprtc): the same bytecode
//                            translated to straight-line code with typed loads - the MI355X-native
//                            counterpart of the reference compiling every query (codegen.py:230-247).
//
// Both share this skeleton, the operator semantics (hs_bin<>) and the dictionary / accumulator code, so
// they produce bit-identical results (tests/test_gpu_kernels.py runs every case through both).
//
// Reference loops replaced: FilterTask.execute tasks.py:167-177, AggregateTask.execute (before_shuffle)
// tasks.py:284-289, fill_aggregators tasks.py:295-310.
//
// Design (HBM-bound; no MFMA - there is no contraction in this workload): every lane streams HS_V = 4
// consecutive rows per step with 16-byte loads, consecutive lanes take consecutive quads (a wave reads
// 1 KiB per f32 column per instruction); the next quad's loads are issued before the current quad is
// evaluated.  Group keys are resolved in a per-workgroup LDS dictionary; every lane owns a private
// accumulator table in LDS laid out [slot][acc][lane] (conflict-free ds_read_b64 / ds_write_b64, no
// atomics), reduced at the end in a fixed order -> bitwise reproducible.
#pragma once

#include "hs_device.h"

#define HS_FUSED_COLS 8 /* numeric column slots preloaded per step */

// 16-byte vector types usable with __builtin_nontemporal_load (HIP's float4 & co. are structs)
typedef float hs_f32x4 __attribute__((ext_vector_type(4)));
typedef int hs_i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long hs_u64x2 __attribute__((ext_vector_type(2)));

// ---- unit combine (fixed-order fold of a unit's workgroup partials + quantisation to the shuffle-file kinds) ----------
// Reference: is_last emission tasks.py:272-278 -> WriteToShufflePartitions.write tasks.py:373 -> io.py:87-94.
// Runs either as its own launch (k_agg_unit, one workgroup per unit) or as the EPILOGUE of the scan kernel: the
// workgroup that finishes a unit's last chunk combines the unit (hs_agg_main_body), so the combine of every unit but
// the last overlaps the scan and the query's tail loses a launch.
struct AggUnitArgs {
    hs_col key;
    hs_agg_spec spec;
    int32_t group_cap;
    int32_t hashed;
    int32_t batch;  // chunks staged through LDS per step (<= HS_UNIT_BATCH, sized to the LDS budget)
    int32_t pad;
    const int64_t* unit_chunk0;
    const uint64_t* part_keys;
    const int64_t* part_rep;
    const uint64_t* part_acc;
    int64_t* out_rep;      // [n_units][GC]
    uint64_t* out_acc;     // [n_units][GC][NA] quantised
    int32_t* out_ngroups;  // [n_units]
    uint32_t* flags;
    // slab emission (hs_agg_partial_slab): rows of unit u dense from slab row u * GC; NULL = the arrays above
    uint8_t* slab;
    const int64_t* unit_ids;
    int64_t order_off, key_off;
    int64_t acc_off[HS_MAX_ACC];
    int32_t key_bytes, pad2;
};
#define HS_UNIT_BATCH 64 /* chunks per LDS batch */

// Chunk partials handed from one workgroup to another INSIDE a launch (fused combine): per-XCD L2s are not coherent
// and a CU's L1 is never refreshed by other CUs' stores, so both sides use agent-scope (sc1, write-through / L1-
// bypassing) accesses for exactly these words (cdna_hip_programming.md section 6 G16, R1 with a counter).  Across a
// launch boundary (HANDOFF false) plain accesses do.
template <bool HANDOFF, typename T>
__device__ __forceinline__ T hs_ld_part(const T* p) {
    if constexpr (HANDOFF) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <typename T>
__device__ __forceinline__ void hs_st_part(T* p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One (unit slot, accumulator) cell over the staged chunks of a batch, in ascending chunk order.  The batch is staged
// CELL-major (x[cell][chunk], a chunk without the key holds the aggregate's identity - a no-op for every kind: +0.0 never
// changes a sum that started from +0.0, MIN / MAX never pass their identity), so the lane reads consecutive LDS words and
// the only dependent chain is the folds themselves; the kind is a template argument (lanes of different kinds take their
// loops one after the other).  (Round 4 history, profiles/r04_scan_stamps_*.txt: the run-time-kind loop through the
// slot map took ~160 ns per chunk - 10 us per 64-chunk batch, most of the combine at sf=1; kind as a template argument
// and eight map / partial reads in flight: 84 ns; this form: the reads are a stream.)
template <int OP, bool IS_INT>
__device__ __forceinline__ void hs_fold_staged_cell(uint64_t* cell, const uint64_t* x, int nb) {
    uint64_t v = *cell;
    int c = 0;
    for (; c + 8 <= nb; c += 8) {
        uint64_t t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = x[c + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) v = hs_acc_fold(OP, IS_INT, v, t[k]);
    }
    for (; c < nb; ++c) v = hs_acc_fold(OP, IS_INT, v, x[c]);
    *cell = v;
}

// Phase 1: every (chunk, slot) entry of the unit is inserted into the unit's dictionary in parallel (the SET of keys
// does not depend on insertion races) and remembers its unit slot.  Phase 2: one lane per (unit slot, accumulator)
// folds that key's chunk partials in ascending chunk order = row order, without barriers.  Chunks are staged through
// LDS in batches.  `lds`: GC * 16 + 2 * GC * NA * 8 + batch * (GC * NA * 8 + GC * 4) bytes.  All lanes of the workgroup.
// `bounds`: the unit's first and one-past-last chunk when the caller has read them already (LDS), else NULL.
template <bool HANDOFF>
__device__ __forceinline__ void hs_agg_unit_body(const AggUnitArgs& A, const int64_t u, uint64_t* lds, const int64_t* bounds = nullptr,
                                                 int64_t* st = nullptr) {
#define HS_UNIT_STAMP(i) do { if (st && threadIdx.x == 0) st[i] = (int64_t)wall_clock64(); } while (0)
    const int GC = A.group_cap, NA = A.spec.n_acc;
    const HsSpecBits sb = hs_spec_bits(A.spec);
    uint64_t* ukeys = lds;                                  // [GC]
    int64_t* ureps = (int64_t*)(lds + GC);                  // [GC]
    uint64_t* uacc = lds + 2 * GC;                          // [GC][NA]
    uint64_t* pacc = uacc + GC * NA;                        // [GC * NA][BATCH + 1] staged chunk partials, cell-major
    const int BATCH = A.batch;
    int* inv = (int*)(pacc + (size_t)(BATCH + 1) * GC * NA);  // [BATCH][GC] chunk slot -> unit slot (or -1)
    __shared__ int s_count;
    const int tid = threadIdx.x, nthr = blockDim.x;
    uint32_t err = 0;
    // the scan's status so far (fused: every chunk of THIS unit has OR-ed its bits in before arriving); asked for here,
    // used when the rows are written - a round trip that used to sit at the very end of the launch
    uint32_t scan_flags = 0;
    if (A.slab && tid == 0) scan_flags = __hip_atomic_load(A.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    for (int i = tid; i < GC; i += nthr) {
        ukeys[i] = HS_EMPTY_KEY;
        ureps[i] = -1;
    }
    for (int i = tid; i < GC * NA; i += nthr) uacc[i] = hs_acc_identity(hs_spec_op(sb, i % NA), hs_spec_int(sb, i % NA));
    if (tid == 0) s_count = 0;
    __syncthreads();
    HS_UNIT_STAMP(8);

    const uint32_t mask = (uint32_t)GC - 1;
    const int64_t cbeg = bounds ? bounds[0] : A.unit_chunk0[u], cend = bounds ? bounds[1] : A.unit_chunk0[u + 1];
    constexpr int HS_UNIT_MLP = 8;  // global loads a lane keeps in flight while staging (one wait per group, not per load)
    const int XS = BATCH + 1;       // cell stride of the staged batch (odd: the folding lanes start in different banks)
    for (int64_t b0 = cbeg; b0 < cend; b0 += BATCH) {
        const int nb = (int)((cend - b0) < BATCH ? (cend - b0) : BATCH);
        const int nent = nb * GC, ncell = nb * GC * NA;
        // every global word of the batch is asked for up front: the first pass's dictionary entries and the first
        // HS_UNIT_MLP partials of every lane travel together (one round trip; a load -> wait -> ds_write loop cost one per
        // element: 6 per batch for Q1, 12 at sf=1 = most of the combine's 30 us there)
        int64_t rep0 = -1;
        uint64_t kw0 = 0;
        if (tid < nent) {
            rep0 = hs_ld_part<HANDOFF>(&A.part_rep[b0 * GC + tid]);
            if (!A.hashed) kw0 = hs_ld_part<HANDOFF>(&A.part_keys[b0 * GC + tid]);
        }
        uint64_t v0[HS_UNIT_MLP];
#pragma unroll
        for (int k = 0; k < HS_UNIT_MLP; ++k) {
            const int i = tid + k * nthr;
            v0[k] = i < ncell ? hs_ld_part<HANDOFF>(&A.part_acc[b0 * GC * NA + i]) : 0ull;
        }
        for (int i = tid; i < nent; i += nthr) inv[i] = -1;  // chunk slot -> unit slot
        for (int i = tid; i < GC * NA * nb; i += nthr) {     // x[cell][chunk] = identity until a chunk brings the key
            const int cellid = i / nb, c = i - cellid * nb;
            pacc[cellid * XS + c] = hs_acc_identity(hs_spec_op(sb, cellid % NA), hs_spec_int(sb, cellid % NA));
        }
        __syncthreads();
        for (int i = tid; i < nent; i += nthr) {
            // both words of the entry in one round trip (the key word is only meaningful next to a valid row id)
            const int64_t rep = i == tid ? rep0 : hs_ld_part<HANDOFF>(&A.part_rep[b0 * GC + i]);
            const uint64_t kw = i == tid ? kw0 : A.hashed ? 0ull : hs_ld_part<HANDOFF>(&A.part_keys[b0 * GC + i]);
            if (rep >= 0) {
                const int m = A.hashed ? hs_dict_upsert_rows(ureps, mask, A.key, hs_key_at(A.key, rep), rep)
                                       : hs_dict_upsert_word(ukeys, ureps, mask, kw, rep);
                if (m < 0) err |= HS_FLAG_DICT_FULL;
                else inv[i] = m;
            }
        }
        __syncthreads();
        if (b0 == cbeg) HS_UNIT_STAMP(9);
        // the partials go to their unit slot's row (keys are distinct within a chunk: one writer per word)
#pragma unroll
        for (int k = 0; k < HS_UNIT_MLP; ++k) {
            const int i = tid + k * nthr;
            if (i < ncell) {
                const int c = i / (GC * NA), r = i - c * (GC * NA), sl = r / NA, a = r - sl * NA;
                const int m = inv[c * GC + sl];
                if (m >= 0) pacc[(m * NA + a) * XS + c] = v0[k];
            }
        }
        for (int i = tid + HS_UNIT_MLP * nthr; i < ncell; i += nthr) {  // (wider tables: the rest, one round trip each)
            const int c = i / (GC * NA), r = i - c * (GC * NA), sl = r / NA, a = r - sl * NA;
            const int m = inv[c * GC + sl];
            if (m >= 0) pacc[(m * NA + a) * XS + c] = hs_ld_part<HANDOFF>(&A.part_acc[b0 * GC * NA + i]);
        }
        __syncthreads();
        if (b0 == cbeg) HS_UNIT_STAMP(10);
        for (int i = tid; i < GC * NA; i += nthr) {
            const int a = i % NA;
            const uint32_t op = hs_spec_op(sb, a);
            const bool is_int = hs_spec_int(sb, a);
            const uint64_t* x = pacc + i * XS;
            if (is_int) {
                if (op == HS_AGG_SUM) hs_fold_staged_cell<HS_AGG_SUM, true>(&uacc[i], x, nb);
                else if (op == HS_AGG_MIN) hs_fold_staged_cell<HS_AGG_MIN, true>(&uacc[i], x, nb);
                else hs_fold_staged_cell<HS_AGG_MAX, true>(&uacc[i], x, nb);
            } else {
                if (op == HS_AGG_SUM) hs_fold_staged_cell<HS_AGG_SUM, false>(&uacc[i], x, nb);
                else if (op == HS_AGG_MIN) hs_fold_staged_cell<HS_AGG_MIN, false>(&uacc[i], x, nb);
                else hs_fold_staged_cell<HS_AGG_MAX, false>(&uacc[i], x, nb);
            }
        }
        __syncthreads();
        if (b0 == cbeg) HS_UNIT_STAMP(11);
    }
    HS_UNIT_STAMP(12);
    if (A.slab) {
        // the unit's groups, dense from slab row u * GC, in the stored kinds (what the reference's shuffle file
        // holds); dense position of a slot = number of occupied slots before it (wave 0, ballot prefix)
        int* dpos = inv;  // [GC]
        if (tid < HS_WAVE) {
            int running = 0;
            for (int base = 0; base < GC; base += HS_WAVE) {
                const int sl = base + tid;
                const bool valid = sl < GC && ureps[sl] >= 0;
                const unsigned long long m = __ballot(valid);
                if (valid) dpos[sl] = running + __popcll(m & ((1ull << tid) - 1));
                running += __popcll(m);
            }
            if (tid == 0) s_count = running;
        }
        __syncthreads();
        const int count = s_count;
        const int64_t row0 = u * GC;
        int64_t* order = (int64_t*)(A.slab + A.order_off) + row0;
        const int64_t uid = A.unit_ids ? A.unit_ids[u] : u;
        for (int i = tid; i < GC; i += nthr) order[i] = i < count ? uid : -1;
        const int kb = A.key_bytes;
        for (int sl = tid; sl < GC; sl += nthr) {
            const int64_t rep = ureps[sl];
            if (rep < 0) continue;
            uint8_t* dst = A.slab + A.key_off + (row0 + dpos[sl]) * kb;
            if (!A.hashed && A.key.kind == HS_STR) {  // a packed string's key word holds its bytes: no trip back to the column
                const uint64_t w = ukeys[sl];
                for (int b = 0; b < kb; ++b) dst[b] = (uint8_t)(w >> (8 * b));
                continue;
            }
            const uint8_t* src = (const uint8_t*)A.key.data + rep * kb;
            for (int b = 0; b < kb; ++b) dst[b] = src[b];
        }
        for (int i = tid; i < GC * NA; i += nthr) {
            const int sl = i / NA, a = i % NA;
            if (ureps[sl] < 0) continue;
            const bool is_int = hs_spec_int(sb, a);
            if (hs_float_identity_left(hs_spec_op(sb, a), is_int, uacc[i])) err |= HS_FLAG_TYPE_ASSERT;
            const uint64_t v = hs_quantise_cell(is_int, uacc[i], err);
            uint8_t* col = A.slab + A.acc_off[a];
            if (is_int) ((int32_t*)col)[row0 + dpos[sl]] = (int32_t)(int64_t)v;
            else ((float*)col)[row0 + dpos[sl]] = (float)hs_u2d(v);
        }
        err |= scan_flags;
        if (err) {
            atomicOr(A.flags, err);
            atomicOr((uint32_t*)A.slab, err);  // slab header: reaches every rank with the rows
        }
        HS_UNIT_STAMP(13);
        return;
    }
    for (int sl = tid; sl < GC; sl += nthr) {
        // representative = smallest row index seen for the slot would need a second pass; any row of the
        // key is equivalent (same key bytes), so keep the one that won the insert
        const int64_t rep = ureps[sl];
        A.out_rep[u * GC + sl] = rep;
        if (rep >= 0) atomicAdd(&s_count, 1);
    }
    for (int i = tid; i < GC * NA; i += nthr) {
        const int a = i % NA;
        if (ureps[i / NA] >= 0 && hs_float_identity_left(hs_spec_op(sb, a), hs_spec_int(sb, a), uacc[i]))
            err |= HS_FLAG_TYPE_ASSERT;
        A.out_acc[u * (int64_t)GC * NA + i] = hs_quantise_cell(hs_spec_int(sb, a), uacc[i], err);
    }
    __syncthreads();
    if (tid == 0) A.out_ngroups[u] = s_count;
    if (err) atomicOr(A.flags, err);
}

struct AggMainArgs {
    HsCols cols;
    hs_program prog;
    hs_agg_spec spec;
    int32_t key_col;
    int32_t group_cap;
    int32_t chunk_rows;
    int32_t pad;
    const hs_chunk* chunks;  // [n_chunks] one descriptor per workgroup
    const int64_t* unit_chunk0;
    int64_t n_units;
    uint64_t* part_keys;  // [n_chunks][GC]
    int64_t* part_rep;    // [n_chunks][GC]
    uint64_t* part_acc;   // [n_chunks][GC][n_acc]
    uint32_t* flags;
    int32_t replicas;     // shared-dictionary tier: accumulator copies per slot (power of two <= 64), else 0
    int32_t pad2;
    // fused unit combine (private-table tier): arrivals[u] counts the finished chunks of unit u (zero before the first
    // launch; the last arriver zeroes it again); NULL = the combine is a separate launch (k_agg_unit)
    uint32_t* unit_arrivals;
    AggUnitArgs unit;
    // shared-dictionary tier with COMPUTED units (the probe side of a join, rows left in place): column slot of a
    // preloaded HS_U8 column holding every row's unit id (< 128; 0xff = the row takes no part), or -1 = units are
    // the row ranges of `chunks`.  The id rides in the top byte of the key word (hs_unit_key), so the workgroup's
    // LDS dictionary holds (unit, key) pairs and the chunk merge files every entry under its own unit's table.
    int32_t unit_col;
    int32_t pad3;
    // computed units: chunk_acc[chunk][unit][geom pad][n_acc] - every chunk leaves its cells HERE with plain stores and
    // a small kernel folds them per (unit, slot): with ONE row range every chunk holds the same few (unit, key)
    // groups, and ~900 workgroups adding into the same 200 global cells at the end of the scan serialised to 3 ms
    uint64_t* chunk_acc;
    // round 3: the join's byte table when cols hold HS_JOIN8_CODE / HS_JOIN8_UNIT columns (the probe runs inside the
    // scan: include/hipspark.h hs_agg_shared_join8); table NULL otherwise
    hs_join8 join;
    // debug (HIPSPARK_SCAN_STAMPS=1; tools/scan_stamps.py): [n_chunks][16] wall_clock64() at the phase boundaries of every
    // workgroup of the private-table scan + the hardware id; NULL otherwise
    int64_t* stamps;
};
#define HS_SCAN_STAMP(i) do { if (A.stamps && threadIdx.x == 0) A.stamps[(int64_t)blockIdx.x * 16 + (i)] = (int64_t)wall_clock64(); } while (0)

// key word of (unit, key): valid for key words that carry their information in the low 56 bits - INTEGER keys
// (32 significant bits), packed strings of a FIXED length <= 6 (the length byte is the same for every row)
__device__ __forceinline__ uint64_t hs_unit_key(uint64_t k, uint32_t unit) {
    return (k & 0x00ffffffffffffffull) | ((uint64_t)unit << 56);
}

// 64-bit wave shuffle-down
__device__ __forceinline__ uint64_t hs_shfl_down64(uint64_t v, int delta) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_down(lo, delta, HS_WAVE);
    hi = __shfl_down(hi, delta, HS_WAVE);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ bool hs_str_preloads(const hs_col& c) {
    return c.kind == HS_STR && (c.fixed_len == 1 || c.fixed_len == 2 || c.fixed_len == 4);
}

// packed key words of a fixed-width (1 / 2 / 4 byte) string column for one row quad
template <int LEN>
__device__ __forceinline__ void hs_load_packed_quad(const hs_col& c, int64_t row0, uint64_t (&cell)[HS_V]) {
    if constexpr (LEN == 1) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>((const uint8_t*)c.data + row0);
        cell[0] = (1ull << 56) | (v & 0xff);
        cell[1] = (1ull << 56) | ((v >> 8) & 0xff);
        cell[2] = (1ull << 56) | ((v >> 16) & 0xff);
        cell[3] = (1ull << 56) | (v >> 24);
    } else if constexpr (LEN == 2) {
        const uint2 v = *reinterpret_cast<const uint2*>((const uint8_t*)c.data + row0 * 2);
        cell[0] = (2ull << 56) | (v.x & 0xffff);
        cell[1] = (2ull << 56) | (v.x >> 16);
        cell[2] = (2ull << 56) | (v.y & 0xffff);
        cell[3] = (2ull << 56) | (v.y >> 16);
    } else {
        const uint4 v = *reinterpret_cast<const uint4*>((const uint8_t*)c.data + row0 * 4);
        cell[0] = (4ull << 56) | v.x;
        cell[1] = (4ull << 56) | v.y;
        cell[2] = (4ull << 56) | v.z;
        cell[3] = (4ull << 56) | v.w;
    }
}

// one row quad of a column, widened to 64-bit cells (run-time kind: the interpreter's loader)
__device__ __forceinline__ void hs_load_quad(const hs_col& c, int64_t row0, uint64_t (&cell)[HS_V]) {
    switch (c.kind) {
        case HS_I32: {
            const int4 v = *reinterpret_cast<const int4*>((const int32_t*)c.data + row0);
            cell[0] = (uint64_t)(int64_t)v.x;
            cell[1] = (uint64_t)(int64_t)v.y;
            cell[2] = (uint64_t)(int64_t)v.z;
            cell[3] = (uint64_t)(int64_t)v.w;
            break;
        }
        case HS_F32: {
            const float4 v = *reinterpret_cast<const float4*>((const float*)c.data + row0);
            cell[0] = hs_d2u((double)v.x);
            cell[1] = hs_d2u((double)v.y);
            cell[2] = hs_d2u((double)v.z);
            cell[3] = hs_d2u((double)v.w);
            break;
        }
        case HS_I64:
        case HS_F64: {
            const ulonglong2 v0 = *reinterpret_cast<const ulonglong2*>((const uint64_t*)c.data + row0);
            const ulonglong2 v1 = *reinterpret_cast<const ulonglong2*>((const uint64_t*)c.data + row0 + 2);
            cell[0] = v0.x;
            cell[1] = v0.y;
            cell[2] = v1.x;
            cell[3] = v1.y;
            break;
        }
        case HS_U8: {
            const uint32_t v = *reinterpret_cast<const uint32_t*>((const uint8_t*)c.data + row0);
            cell[0] = v & 0xff;
            cell[1] = (v >> 8) & 0xff;
            cell[2] = (v >> 16) & 0xff;
            cell[3] = v >> 24;
            break;
        }
        case HS_STR:
            // only the GROUP BY column is preloaded, and only when it packs with a power-of-two width
            if (c.fixed_len == 1) hs_load_packed_quad<1>(c, row0, cell);
            else if (c.fixed_len == 2) hs_load_packed_quad<2>(c, row0, cell);
            else if (c.fixed_len == 4) hs_load_packed_quad<4>(c, row0, cell);
            break;
        default: break;
    }
}

// Per-lane state of the fused kernel: liveness of the quad's rows, their group slots, the workgroup's
// dictionary and the lane's private accumulator table.
struct AggCtx {
    int64_t row0;
    bool alive[HS_V];
    int slot[HS_V];
    uint64_t* dkeys;
    int64_t* dreps;
    uint64_t* tbl;
    uint8_t* dmap;  // LDS [256]: slot of a one-byte key, 0xff = not seen yet by this workgroup
    uint32_t tid, nthr, mask;
    int32_t n_acc;
    uint32_t err;
    HsSpecBits sb;  // the aggregate description in two scalars (hs_spec_bits)

    // One-byte string keys (TPC-H flags): the byte indexes a 256-entry LDS map straight to its slot - one
    // ds_read_u8 instead of hash + probe loop; the dictionary is only walked the first time a lane meets a byte.
    // Lanes racing on dmap[b] all store the slot the dictionary gave that key, so any order is fine.
    __device__ __forceinline__ int find_byte(const hs_col& key_col, uint64_t k, int64_t row, bool& live) {
        if (mask >= 255u) return find<false>(key_col, k, row, live);  // slots would not fit the map's byte
        int s = 0;
        if (live) {
            const uint32_t b = (uint32_t)k & 0xffu;
            s = dmap[b];
            if (s == 0xff) {
                s = hs_dict_upsert_word(dkeys, dreps, mask, k, row);
                if (s < 0) {
                    err |= HS_FLAG_DICT_FULL;
                    live = false;
                    s = 0;
                } else {
                    dmap[b] = (uint8_t)s;
                }
            }
        }
        return s;
    }

    // slot of key word `k` held by `row`; marks the row dead and flags overflow when the table is full
    template <bool HASHED>
    __device__ __forceinline__ int find(const hs_col& key_col, uint64_t k, int64_t row, bool& live) {
        int s = 0;
        if (live) {
            if constexpr (HASHED) s = hs_dict_upsert_rows(dreps, mask, key_col, k, row);
            else s = hs_dict_upsert_word(dkeys, dreps, mask, k, row);
            if (s < 0) {
                err |= HS_FLAG_DICT_FULL;
                live = false;
                s = 0;
            }
        }
        return s;
    }

    // fold with run-time aggregate description (interpreter)
    __device__ __forceinline__ void fold(const hs_agg_spec& spec, uint32_t a, int s, bool live, uint64_t x) {
        if (live) {
            const uint32_t idx = ((uint32_t)s * (uint32_t)n_acc + a) * nthr + tid;
            (void)spec;
            tbl[idx] = hs_acc_fold(hs_spec_op(sb, a), hs_spec_int(sb, a), tbl[idx], x);
        }
    }

    // fold with compile-time aggregate description (JIT): dead rows fold the identity into slot 0,
    // which leaves every accumulator unchanged, so there is no branch
    template <int NA, int A, int OP, bool IS_INT>
    __device__ __forceinline__ void fold_c(int s, bool live, uint64_t x) {
        const uint32_t idx = ((uint32_t)(live ? s : 0) * (uint32_t)NA + (uint32_t)A) * nthr + tid;
        const uint64_t v = live ? x : hs_acc_identity(OP, IS_INT);
        tbl[idx] = hs_acc_fold(OP, IS_INT, tbl[idx], v);
    }
};

// ---- the interpreter as a Prog ------------------------------------------------------------------------
template <bool HASHED_, int D>
struct InterpProg {
    static constexpr bool HASHED = HASHED_;
    static constexpr bool TWO_STAGE = false;  // (compiled programs with the join's probe inside load in two stages)
    static constexpr bool STATIC_SPEC = false;  // (compiled programs know their aggregates' kinds at compile time)
    static constexpr int NA = 0;
    static __device__ constexpr uint32_t acc_op(int) { return 0; }
    static __device__ constexpr bool acc_int(int) { return false; }
    struct Cells {
        uint64_t cell[HS_FUSED_COLS][HS_V];
    };

    static __device__ __forceinline__ void load(const AggMainArgs& A, int64_t base, Cells& x) {
        const int ncols = A.cols.n < HS_FUSED_COLS ? A.cols.n : HS_FUSED_COLS;
#pragma unroll
        for (int c = 0; c < HS_FUSED_COLS; ++c) {
            if (c < ncols) {
                const hs_col& col = A.cols.c[c];
                if (col.kind != HS_STR || (c == A.key_col && hs_str_preloads(col))) hs_load_quad(col, base, x.cell[c]);
            }
        }
    }

    template <class Ctx>
    struct Sink {
        const AggMainArgs& A;
        const Cells& x;
        Ctx& ctx;
        __device__ __forceinline__ Sink(const AggMainArgs& a, const Cells& c, Ctx& k) : A(a), x(c), ctx(k) {}
        __device__ __forceinline__ void load(uint32_t s, uint64_t (&dst)[HS_V]) const {
            switch (s) {
#define HS_CASE(K)                                                          \
    case K:                                                                 \
        _Pragma("unroll") for (int j = 0; j < HS_V; ++j) dst[j] = x.cell[K][j]; \
        break;
                HS_CASE(0) HS_CASE(1) HS_CASE(2) HS_CASE(3) HS_CASE(4) HS_CASE(5) HS_CASE(6) HS_CASE(7)
#undef HS_CASE
                default: break;
            }
        }
        __device__ __forceinline__ uint64_t load(uint32_t s, int j) const {
            uint64_t tmp[HS_V];
            load(s, tmp);
            return tmp[j];
        }
        __device__ __forceinline__ bool live(int j) const { return ctx.alive[j]; }
        __device__ __forceinline__ int64_t row(int j) const { return ctx.row0 + j; }
        __device__ __forceinline__ void filter(int j, bool keep) { ctx.alive[j] = ctx.alive[j] && keep; }
        __device__ __forceinline__ void out(uint32_t, int, uint64_t) {}
        __device__ __forceinline__ void key() {
            const hs_col& kc = A.cols.c[A.key_col];
            uint64_t kcell[HS_V];
            const bool pre = (kc.kind != HS_STR) || hs_str_preloads(kc);
            if (pre) load((uint32_t)A.key_col, kcell);
            uint64_t ucell[HS_V] = {0, 0, 0, 0};
            if (A.unit_col >= 0) load((uint32_t)A.unit_col, ucell);
#pragma unroll
            for (int j = 0; j < HS_V; ++j) {
                uint64_t k = 0;
                if (A.unit_col >= 0 && ucell[j] == 0xffull) ctx.alive[j] = false;  // e.g. a probe row without a match
                if (ctx.alive[j]) {
                    if (kc.kind == HS_STR) k = pre ? kcell[j] : hs_key_at(kc, ctx.row0 + j);
                    else k = hs_key_from_cell(kc.kind, kcell[j]);
                    if (A.unit_col >= 0) k = hs_unit_key(k, (uint32_t)ucell[j]);
                }
                if (!HASHED && kc.kind == HS_STR && kc.fixed_len == 1 && A.unit_col < 0)
                    ctx.slot[j] = ctx.find_byte(kc, k, ctx.row0 + j, ctx.alive[j]);
                else ctx.slot[j] = ctx.template find<HASHED>(kc, k, ctx.row0 + j, ctx.alive[j]);
            }
        }
        __device__ __forceinline__ void agg(uint32_t a, int j, uint64_t v) {
            ctx.fold(A.spec, a, ctx.slot[j], ctx.alive[j], v);
        }
    };

    template <class Ctx>
    static __device__ __forceinline__ void run(const AggMainArgs& A, const Cells& x, Ctx& ctx) {
        Sink<Ctx> sink(A, x, ctx);
        const hs_program& P = A.prog;
        uint64_t st[D][HS_V];
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int j = 0; j < HS_V; ++j) st[d][j] = 0;
        for (uint32_t pc = 0; pc < P.n_ins; ++pc) {
            const uint64_t w = P.ins[pc];
            const uint32_t sp = hs_ins_sp(w);
            if (hs_ins_op(w) == HS_OP_LD) {  // LD moves the whole quad with one slot switch
                uint64_t tmp[HS_V];
                sink.load(hs_ins_a(w), tmp);
                switch (sp) {
#define HS_PUSH(K)                                                            \
    case K:                                                                   \
        if constexpr (K < D) {                                                \
            _Pragma("unroll") for (int j = 0; j < HS_V; ++j) st[K][j] = tmp[j]; \
        }                                                                     \
        break;
                    HS_PUSH(0) HS_PUSH(1) HS_PUSH(2) HS_PUSH(3) HS_PUSH(4) HS_PUSH(5) HS_PUSH(6) HS_PUSH(7)
#undef HS_PUSH
                    default: ctx.err |= HS_FLAG_BAD_PROGRAM; break;
                }
                continue;
            }
            switch (sp) {
                case 0: hs_exec_at<0, D, HS_V>(w, P, A.cols, st, sink, ctx.err); break;
                case 1: hs_exec_at<1, D, HS_V>(w, P, A.cols, st, sink, ctx.err); break;
                case 2: hs_exec_at<2, D, HS_V>(w, P, A.cols, st, sink, ctx.err); break;
                case 3: hs_exec_at<3, D, HS_V>(w, P, A.cols, st, sink, ctx.err); break;
                case 4: hs_exec_at<4, D, HS_V>(w, P, A.cols, st, sink, ctx.err); break;
                default:
                    if constexpr (D > 4) {
                        switch (sp) {
                            case 5: hs_exec_at<5, D, HS_V>(w, P, A.cols, st, sink, ctx.err); break;
                            case 6: hs_exec_at<6, D, HS_V>(w, P, A.cols, st, sink, ctx.err); break;
                            case 7: hs_exec_at<7, D, HS_V>(w, P, A.cols, st, sink, ctx.err); break;
                            case 8: hs_exec_at<8, D, HS_V>(w, P, A.cols, st, sink, ctx.err); break;
                            default: ctx.err |= HS_FLAG_BAD_PROGRAM; break;
                        }
                    } else {
                        ctx.err |= HS_FLAG_BAD_PROGRAM;
                    }
                    break;
            }
        }
    }
};

// End of a chunk: its partials (agent-scope stores above) are drained, then ONE lane counts the arrival; the workgroup
// that brings a unit's count to its number of chunks combines the unit right here (all its lanes; `lds` = the kernel's
// dynamic LDS block, free again at this point).  Every workgroup arrives - also one that left early - so the counters
// are back at zero when the launch ends.
__device__ __forceinline__ void hs_agg_main_arrive(const AggMainArgs& A, const int64_t unit, uint64_t* lds) {
    if (!A.unit_arrivals) return;  // wave-uniform (kernel argument)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave, before the barrier
    __syncthreads();                                  // ... which also frees the LDS tables for the combine
    __shared__ int s_last;
    __shared__ int64_t s_bounds[2];
    if (threadIdx.x == 0) {
        const int64_t cb = A.unit_chunk0[unit], ce = A.unit_chunk0[unit + 1];
        s_bounds[0] = cb;
        s_bounds[1] = ce;
        const uint32_t need = (uint32_t)(ce - cb);
        const uint32_t prev = __hip_atomic_fetch_add(&A.unit_arrivals[unit], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev + 1 == need;
        if (s_last) __hip_atomic_store(&A.unit_arrivals[unit], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    HS_SCAN_STAMP(5);
    if (s_last) hs_agg_unit_body<true>(A.unit, unit, lds, s_bounds, A.stamps ? A.stamps + (int64_t)blockIdx.x * 16 : nullptr);
}

// ---- the kernel body ---------------------------------------------------------------------------------
template <class Prog>
__device__ __forceinline__ void hs_agg_main_body(const AggMainArgs& A) {
    extern __shared__ __align__(16) uint64_t hs_lds[];
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const int GC = A.group_cap;
    const int NA = A.spec.n_acc;
    uint64_t* dkeys = hs_lds;
    int64_t* dreps = (int64_t*)(hs_lds + GC);
    uint64_t* tbl = hs_lds + 2 * GC;

    // this workgroup's row range: one 32-byte descriptor (a wave-uniform scalar load)
    const int64_t chunk = blockIdx.x;
    HS_SCAN_STAMP(0);
    if (A.stamps && threadIdx.x == 0)  // XCC_ID (hwreg 20) << 32 | HW_ID (hwreg 4): which XCD / SE / CU ran the chunk
        A.stamps[chunk * 16 + 7] = ((int64_t)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 20) << 32) |
                                  (int64_t)(uint32_t)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);
    const hs_chunk desc = A.chunks[chunk];
    const int64_t us = desc.unit_begin, c0 = desc.row_begin, c1 = desc.row_end;
    // the first quad's loads go out first of all: their latency hides the early-exit check and the LDS initialisation
    const int64_t stride = (int64_t)nthr * HS_V;
    int64_t base = c0 + (int64_t)tid * HS_V;
    typename Prog::Cells cur, nxt;
    if (base < c1) Prog::load(A, base, nxt);
    const HsSpecBits sb = hs_spec_bits(A.spec);

    // a run that has already overflowed a dictionary is going to be repeated with larger tables: later rounds of
    // workgroups only mark their chunk empty and leave
    // (the decision must be the SAME for every lane of the workgroup - one lane reads the flag, LDS hands it round:
    // lanes leaving on their own would let the others run on half-initialised tables and emit garbage row ids)
    __shared__ uint32_t s_overflowed;
    if (threadIdx.x == 0) s_overflowed = __hip_atomic_load(A.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & HS_FLAG_DICT_FULL;
    // the tables are initialised on this side of the barrier the check needs anyway (a workgroup that leaves does not care)
    for (int i = tid; i < GC; i += nthr) {
        dkeys[i] = HS_EMPTY_KEY;
        dreps[i] = -1;
    }
    __shared__ uint8_t s_dmap[256];
    for (int i = tid; i < 256; i += nthr) s_dmap[i] = 0xff;
    for (int cellid = 0; cellid < GC * NA; ++cellid)
        tbl[(uint32_t)cellid * nthr + tid] = hs_acc_identity(hs_spec_op(sb, cellid % NA), hs_spec_int(sb, cellid % NA));
    __syncthreads();
    HS_SCAN_STAMP(1);
    if (s_overflowed) {
        for (int i = threadIdx.x; i < A.group_cap; i += blockDim.x) {
            hs_st_part(&A.part_keys[chunk * A.group_cap + i], HS_EMPTY_KEY);
            hs_st_part(&A.part_rep[chunk * A.group_cap + i], (int64_t)-1);
        }
    } else {  // (one call site of the arrival / unit combine below: it is inlined, and long)
    AggCtx ctx;
    ctx.dkeys = dkeys;
    ctx.dreps = dreps;
    ctx.tbl = tbl;
    ctx.dmap = s_dmap;
    ctx.tid = tid;
    ctx.nthr = nthr;
    ctx.mask = (uint32_t)GC - 1;
    ctx.n_acc = NA;
    ctx.err = 0;
    ctx.sb = sb;
    HS_SCAN_STAMP(2);

    // software pipeline: the loads of step i+1 are in flight while step i is evaluated
    while (base < c1) {
        cur = nxt;
        const int64_t next_base = base + stride;
        if (next_base < c1) Prog::load(A, next_base, nxt);
        ctx.row0 = base;
#pragma unroll
        for (int j = 0; j < HS_V; ++j) {
            const int64_t r = base + j;
            ctx.alive[j] = (r >= us) && (r < c1);
            ctx.slot[j] = 0;
        }
        Prog::run(A, cur, ctx);
        base = next_base;
    }
    HS_SCAN_STAMP(3);
    __syncthreads();
    HS_SCAN_STAMP(14);

    // Fixed-order reduction of the private tables.  A wave owns cells wave, wave+nwaves, ...; it folds
    // each cell's lanes (stride 64), then a shuffle tree.  HS_RED cells are reduced together so that
    // their shuffle chains (each ~6 dependent cross-lane hops) overlap instead of running back to back.
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid / HS_WAVE)), lane = tid % HS_WAVE, nwaves = nthr / HS_WAVE;
    if constexpr (Prog::STATIC_SPEC && (Prog::NA > 0)) {
        // Compiled programs: a wave takes whole SLOTS, and a slot's accumulators are a compile-time list - the folds below
        // are straight-line code of the right kind, all of a slot's cross-lane reads are in flight together.  (With the
        // kinds as run-time values every fold was a tree of branches: ~5 us of this epilogue on an idle chip,
        // profiles/r04_scan_stamps_*.txt.)  Per cell the order of the folds is the generic form's: same bits.
        constexpr int SNA = Prog::NA;
        for (uint32_t sl = wave; sl < (uint32_t)GC; sl += nwaves) {
            if (__builtin_amdgcn_readfirstlane((int)(dreps[sl] >= 0)) == 0) continue;  // wave-uniform
            uint64_t v[SNA], t[SNA];
#pragma unroll
            for (int a = 0; a < SNA; ++a) v[a] = hs_acc_identity(Prog::acc_op(a), Prog::acc_int(a));
            for (uint32_t t0 = lane; t0 < nthr; t0 += HS_WAVE) {
#pragma unroll
                for (int a = 0; a < SNA; ++a) t[a] = tbl[(sl * (uint32_t)SNA + (uint32_t)a) * nthr + t0];
#pragma unroll
                for (int a = 0; a < SNA; ++a) v[a] = hs_acc_fold(Prog::acc_op(a), Prog::acc_int(a), v[a], t[a]);
            }
#pragma unroll
            for (int d = HS_WAVE / 2; d >= 1; d >>= 1) {
#pragma unroll
                for (int a = 0; a < SNA; ++a) t[a] = hs_shfl_down64(v[a], d);
#pragma unroll
                for (int a = 0; a < SNA; ++a) v[a] = hs_acc_fold(Prog::acc_op(a), Prog::acc_int(a), v[a], t[a]);
            }
            if (lane == 0) {
#pragma unroll
                for (int a = 0; a < SNA; ++a) hs_st_part(&A.part_acc[((int64_t)chunk * GC + sl) * SNA + a], v[a]);
            }
        }
    } else {
    // Everything that selects code here is wave-uniform and is kept in SGPRs on purpose (readfirstlane): with the wave
    // number in a VGPR the aggregate kinds were per-lane values and every fold became a tree of exec-mask branches.
    constexpr int HS_RED = 6;
    const uint32_t ncells = (uint32_t)(GC * NA);
    for (uint32_t c0id = wave; c0id < ncells; c0id += nwaves * HS_RED) {
        uint64_t v[HS_RED], t[HS_RED];
        uint32_t ops[HS_RED];
        bool ints[HS_RED], on[HS_RED];
#pragma unroll
        for (int k = 0; k < HS_RED; ++k) {
            const uint32_t cellid = c0id + (uint32_t)k * nwaves;
            on[k] = cellid < ncells && __builtin_amdgcn_readfirstlane((int)(dreps[cellid < ncells ? cellid / NA : 0] >= 0)) != 0;
            const uint32_t a = on[k] ? cellid % NA : 0;
            ops[k] = hs_spec_op(sb, a);
            ints[k] = hs_spec_int(sb, a);
            v[k] = hs_acc_identity(ops[k], ints[k]);
        }
        for (uint32_t t0 = lane; t0 < nthr; t0 += HS_WAVE) {
#pragma unroll
            for (int k = 0; k < HS_RED; ++k) t[k] = on[k] ? tbl[(c0id + (uint32_t)k * nwaves) * nthr + t0] : 0ull;
#pragma unroll
            for (int k = 0; k < HS_RED; ++k)
                if (on[k]) v[k] = hs_acc_fold(ops[k], ints[k], v[k], t[k]);
        }
#pragma unroll
        for (int d = HS_WAVE / 2; d >= 1; d >>= 1) {
#pragma unroll
            for (int k = 0; k < HS_RED; ++k) t[k] = hs_shfl_down64(v[k], d);
#pragma unroll
            for (int k = 0; k < HS_RED; ++k) v[k] = hs_acc_fold(ops[k], ints[k], v[k], t[k]);
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < HS_RED; ++k) {
                const uint32_t cellid = c0id + (uint32_t)k * nwaves;
                if (on[k]) hs_st_part(&A.part_acc[((int64_t)chunk * GC + cellid / NA) * NA + cellid % NA], v[k]);
            }
        }
    }
    }
    for (int i = tid; i < GC; i += nthr) {
        hs_st_part(&A.part_keys[chunk * GC + i], dkeys[i]);
        hs_st_part(&A.part_rep[chunk * GC + i], dreps[i]);
    }
    if (ctx.err) atomicOr(A.flags, ctx.err);
    }
    HS_SCAN_STAMP(4);
    hs_agg_main_arrive(A, desc.unit, hs_lds);
    HS_SCAN_STAMP(6);
}

// ======================================================================================================
// Shared-dictionary tier: GROUP BY with tens to thousands of groups per unit.
//
// Private per-lane tables (above) stop paying once group_cap x n_acc x 8 bytes per LANE no longer leaves room for
// enough waves.  Here the workgroup (1024 lanes, one per CU) owns ONE table in LDS - key words, representative
// rows, n_acc accumulators per slot - and every row updates it with LDS atomics (ds_add_f64 / ds_add_u64 /
// ds_min / ds_max: the LDS does ~10^4 x the atomic rate of HBM).  At the end of its chunk the workgroup merges
// the occupied slots into the UNIT's table in global memory (dictionary upsert + one global atomic per cell:
// chunks x groups of them, negligible next to the rows).  fp64 additions of a group happen in hardware order,
// so this tier is not bitwise reproducible run to run: results stay within ~1e-13 relative of the reference's
// sequential sum (north_star: 1e-9) and equal after f32 rounding except at a rounding boundary (<= 1 ulp).
// ======================================================================================================
template <int OP, bool IS_INT>
__device__ __forceinline__ void hs_atomic_fold_lds(uint64_t* cell, uint64_t x) {
    if constexpr (IS_INT) {
        if constexpr (OP == HS_AGG_SUM) atomicAdd((unsigned long long*)cell, (unsigned long long)x);
        else if constexpr (OP == HS_AGG_MIN) atomicMin((long long*)cell, (long long)x);
        else atomicMax((long long*)cell, (long long)x);
    } else {
        if constexpr (OP == HS_AGG_SUM) atomicAdd((double*)cell, hs_u2d(x));
        else if constexpr (OP == HS_AGG_MIN) atomicMin((double*)cell, hs_u2d(x));
        else atomicMax((double*)cell, hs_u2d(x));
    }
}
template <int OP, bool IS_INT>
__device__ __forceinline__ void hs_atomic_fold_global(uint64_t* cell, uint64_t x) {
    if constexpr (IS_INT) {
        if constexpr (OP == HS_AGG_SUM) atomicAdd((unsigned long long*)cell, (unsigned long long)x);
        else if constexpr (OP == HS_AGG_MIN) atomicMin((long long*)cell, (long long)x);
        else atomicMax((long long*)cell, (long long)x);
    } else {
        if constexpr (OP == HS_AGG_SUM) unsafeAtomicAdd((double*)cell, hs_u2d(x));
        else if constexpr (OP == HS_AGG_MIN) unsafeAtomicMin((double*)cell, hs_u2d(x));
        else unsafeAtomicMax((double*)cell, hs_u2d(x));
    }
}
#define HS_DISPATCH_FOLD(FN, op, is_int, cell, x)                                                         \
    do {                                                                                                  \
        if (is_int) {                                                                                     \
            if ((op) == HS_AGG_SUM) FN<HS_AGG_SUM, true>(cell, x);                                        \
            else if ((op) == HS_AGG_MIN) FN<HS_AGG_MIN, true>(cell, x);                                   \
            else FN<HS_AGG_MAX, true>(cell, x);                                                           \
        } else {                                                                                          \
            if ((op) == HS_AGG_SUM) FN<HS_AGG_SUM, false>(cell, x);                                       \
            else if ((op) == HS_AGG_MIN) FN<HS_AGG_MIN, false>(cell, x);                                  \
            else FN<HS_AGG_MAX, false>(cell, x);                                                          \
        }                                                                                                 \
    } while (0)

struct SharedCtx {
    int64_t row0;
    bool alive[HS_V];
    int slot[HS_V];
    uint64_t* dkeys;
    int64_t* dreps;
    uint64_t* acc;  // [slot][replica][n_acc]; a lane uses replica lane % replicas: with few groups the lanes of a
                    // wave would otherwise queue up on the same LDS words
    uint32_t rep, nrep;
    int* full;  // LDS: set once the table has overflowed - a probe of a full table walks every slot (read and
                // written with relaxed atomics: a volatile access would leave the LDS address space, see hs_device.h)
    int* count;  // LDS: keys in the table so far; past `limit` the table counts as overflowed, so that the run is
                 // repeated with a larger one: the lanes of a wave probe in lockstep, i.e. every lookup costs the
                 // LONGEST probe sequence among 64 (at 78 % load: ~17 probes per row; at 25 %: ~2)
    int32_t limit;
    uint32_t mask;
    int32_t n_acc;
    uint32_t err;

    template <bool HASHED>
    __device__ __forceinline__ int find(const hs_col& key_col, uint64_t k, int64_t row, bool& live) {
        int s = 0;
        if (live) {
            if (__hip_atomic_load(full, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                s = -1;
            } else {
                bool inserted;
                // double hashing: the probe stride comes from other hash bits than the start slot, so a crowded
                // table has no clusters for one unlucky lane to walk while the other 63 wait
                const uint32_t h = hs_slot_hash_strong(k);
                const uint32_t stride = (h >> 16) | 1u;
                if constexpr (HASHED) s = hs_dict_upsert_rows_at(dreps, mask, key_col, k, row, h, inserted, stride);
                else s = hs_dict_upsert_word_at(dkeys, dreps, mask, k, row, h, inserted, stride);
                if (s < 0 || (inserted && atomicAdd(count, 1) >= limit)) {
                    __hip_atomic_store(full, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    s = -1;
                }
            }
            if (s < 0) {
                err |= HS_FLAG_DICT_FULL;
                live = false;
                s = 0;
            }
        }
        return s;
    }
    __device__ __forceinline__ int find_byte(const hs_col& key_col, uint64_t k, int64_t row, bool& live) {
        return find<false>(key_col, k, row, live);
    }
    __device__ __forceinline__ void fold(const hs_agg_spec& spec, uint32_t a, int s, bool live, uint64_t x) {
        if (live)
            HS_DISPATCH_FOLD(hs_atomic_fold_lds, spec.op[a], spec.is_int[a] != 0,
                             &acc[((uint32_t)s * nrep + rep) * (uint32_t)n_acc + a], x);
    }
    template <int NA, int A, int OP, bool IS_INT>
    __device__ __forceinline__ void fold_c(int s, bool live, uint64_t x) {
        if (live) hs_atomic_fold_lds<OP, IS_INT>(&acc[((uint32_t)s * nrep + rep) * (uint32_t)NA + (uint32_t)A], x);
    }
};

// the unit's table in global memory: same open addressing as the LDS dictionaries, claimed with global CAS
__device__ __forceinline__ int64_t hs_unit_upsert(uint64_t* keys, int64_t* reps, uint32_t mask, bool hashed,
                                                  const hs_col& key_col, uint64_t k, int64_t row) {
    uint32_t h = hs_slot_hash_strong(k) & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        if (hashed) {
            long long cur = __hip_atomic_load(&reps[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur < 0) {
                cur = (long long)atomicCAS((unsigned long long*)&reps[h], (unsigned long long)(-1ll), (unsigned long long)row);
                if (cur < 0) return (int64_t)h;
            }
            if (hs_rows_equal(key_col, (int64_t)cur, row)) return (int64_t)h;
        } else {
            uint64_t cur = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == HS_EMPTY_KEY) {
                cur = atomicCAS((unsigned long long*)&keys[h], (unsigned long long)HS_EMPTY_KEY, (unsigned long long)k);
                if (cur == HS_EMPTY_KEY) {
                    reps[h] = row;  // read only after the kernel
                    return (int64_t)h;
                }
            }
            if (cur == k) return (int64_t)h;
        }
        h = (h + 1) & mask;
    }
    return -1;
}

// AggMainArgs as used by this tier: group_cap = slots of the LDS table, pad = slots of a unit's global table,
// part_keys / part_rep / part_acc = the unit tables [n_units][pad] (keys EMPTY, reps -1, cells = identities)
template <class Prog>
__device__ __forceinline__ void hs_agg_shared_body(const AggMainArgs& A) {
    extern __shared__ __align__(16) uint64_t hs_lds[];
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const int GC = A.group_cap;
    const int NA = A.spec.n_acc;
    uint64_t* dkeys = hs_lds;
    int64_t* dreps = (int64_t*)(hs_lds + GC);
    uint64_t* acc = hs_lds + 2 * GC;

    const hs_chunk desc = A.chunks[blockIdx.x];
    const int64_t us = desc.unit_begin, c0 = desc.row_begin, c1 = desc.row_end;
    // a run that has already overflowed a dictionary is going to be repeated with larger tables: stop early
    // (one lane reads the flag and LDS hands it round: the decision must be uniform across the workgroup)
    __shared__ uint32_t s_overflowed;
    if (threadIdx.x == 0) s_overflowed = __hip_atomic_load(A.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & HS_FLAG_DICT_FULL;
    __syncthreads();
    if (s_overflowed) return;

    const int64_t stride = (int64_t)nthr * HS_V;
    int64_t base = c0 + (int64_t)tid * HS_V;
    typename Prog::Cells cur, nxt, far;
    if constexpr (Prog::TWO_STAGE) {
        // The join's probe is a DEPENDENT load (probe key -> table byte).  Keys travel two steps ahead of the rows being
        // folded, table bytes (with the step's other columns) one step ahead: no wave waits for a load it has just
        // issued.  (One stage - keys and bytes in the same prefetch - left the waves parked 60 % of their cycles.)
        if (base < c1) Prog::load_keys(A, base, nxt);
        if (base + stride < c1) Prog::load_keys(A, base + stride, far);
    }
    if (base < c1) Prog::load(A, base, nxt);
    for (int i = tid; i < GC; i += nthr) {
        dkeys[i] = HS_EMPTY_KEY;
        dreps[i] = -1;
    }
    const int R = A.replicas > 0 ? A.replicas : 1;
    const HsSpecBits sb = hs_spec_bits(A.spec);
    for (int i = tid; i < GC * R * NA; i += nthr) acc[i] = hs_acc_identity(hs_spec_op(sb, i % NA), hs_spec_int(sb, i % NA));
    __shared__ int s_full, s_count;
    if (tid == 0) {
        s_full = 0;
        s_count = 0;
    }
    __syncthreads();

    SharedCtx ctx;
    ctx.dkeys = dkeys;
    ctx.dreps = dreps;
    ctx.acc = acc;
    ctx.rep = tid & (uint32_t)(R - 1);
    ctx.nrep = (uint32_t)R;
    ctx.full = &s_full;
    ctx.count = &s_count;
    ctx.limit = A.pad2 > 0 ? A.pad2 : GC;
    ctx.mask = (uint32_t)GC - 1;
    ctx.n_acc = NA;
    ctx.err = 0;
    while (base < c1) {
        cur = nxt;
        const int64_t next_base = base + stride;
        if constexpr (Prog::TWO_STAGE) {
            if (next_base < c1) {
                Prog::take_keys(nxt, far);
                Prog::load(A, next_base, nxt);
                if (next_base + stride < c1) Prog::load_keys(A, next_base + stride, far);
            }
        } else {
            if (next_base < c1) Prog::load(A, next_base, nxt);
        }
        ctx.row0 = base;
#pragma unroll
        for (int j = 0; j < HS_V; ++j) {
            const int64_t r = base + j;
            ctx.alive[j] = (r >= us) && (r < c1);
            ctx.slot[j] = 0;
        }
        Prog::run(A, cur, ctx);
        base = next_base;
        if (__hip_atomic_load(&s_full, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;  // overflowed: the run is repeated with a larger table
    }
    __syncthreads();
    if (s_full) {
        if (tid == 0) atomicOr(A.flags, HS_FLAG_DICT_FULL);
        return;
    }

    // merge this chunk's groups into the unit's table
    const int UC = A.pad;
    const hs_col& kc = A.cols.c[A.key_col];
    if (A.chunk_acc) {  // this chunk's cells start as identities (most (unit, slot) pairs stay that way)
        uint64_t* mine = A.chunk_acc + (int64_t)blockIdx.x * A.n_units * (int64_t)UC * NA;
        const int cells = (int)A.n_units * UC * NA;
        for (int i = tid; i < cells; i += nthr) mine[i] = hs_acc_identity(hs_spec_op(sb, i % NA), hs_spec_int(sb, i % NA));
        __syncthreads();  // (drains the stores: a cell overwritten below must not be overtaken by its identity)
    }
    for (int sl = tid; sl < GC; sl += nthr) {
        const int64_t rep = dreps[sl];
        if (rep < 0) continue;
        const uint64_t k = Prog::HASHED ? hs_key_at(kc, rep) : dkeys[sl];
        // computed units: the entry says which unit it belongs to (never with hashed keys: the host refuses those)
        const int64_t unit = A.unit_col >= 0 ? (int64_t)(k >> 56) : desc.unit;
        uint64_t* ukeys = A.part_keys + unit * UC;
        int64_t* ureps = A.part_rep + unit * UC;
        uint64_t* uacc = A.part_acc + unit * (int64_t)UC * NA;
        const int64_t u = hs_unit_upsert(ukeys, ureps, (uint32_t)UC - 1, Prog::HASHED, kc, k, rep);
        if (u < 0) {
            ctx.err |= HS_FLAG_DICT_FULL;
            continue;
        }
        uint64_t* mine = A.chunk_acc ? A.chunk_acc + ((int64_t)blockIdx.x * A.n_units + unit) * (int64_t)UC * NA : nullptr;
        for (int a = 0; a < NA; ++a) {
            const uint32_t op = hs_spec_op(sb, a);
            const bool is_int = hs_spec_int(sb, a);
            uint64_t v = acc[(sl * R) * NA + a];
            for (int r = 1; r < R; ++r) v = hs_acc_fold(op, is_int, v, acc[(sl * R + r) * NA + a]);  // replicas, in order
            if (mine) mine[u * NA + a] = v;  // this chunk's own cell: no contention (k_agg_shared_fold_chunks adds them up)
            else HS_DISPATCH_FOLD(hs_atomic_fold_global, op, is_int, &uacc[u * NA + a], v);
        }
    }
    if (ctx.err) atomicOr(A.flags, ctx.err);
}

// ---- arguments of the expression-evaluation kernels (hs_eval: interpreter k_eval and the compiled k_eval_jit) ----
struct EvalArgs {
    HsCols cols;
    hs_program prog;
    const int64_t* sel;
    int64_t nrows;
    const int64_t* nrows_dev;
    void* outs[HS_MAX_OUTS];
    int32_t out_kinds[HS_MAX_OUTS];
    uint32_t* flags;
};
