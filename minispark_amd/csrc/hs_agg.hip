// hs_agg.hip - hash group-by / aggregate on gfx950.
//
//   k_agg_main   scan + WHERE + aggregate-argument evaluation + per-workgroup partial aggregate: body in
//                hs_agg_kernel.h (interpreter instantiated here, JIT-compiled programs in hs_jit.cpp)
//   k_agg_unit   fixed-order combine of a unit's workgroup partials + quantisation to the shuffle-file
//                types (reference: is_last emission tasks.py:272-278 -> WriteToShufflePartitions.write
//                tasks.py:373 -> io.py:87-94)
//   k_agg_pack   dense partial rows (the reference's shuffle file content)
//   k_agg_merge_small  final merge of partial rows in unit order (reference: after-shuffle branch
//                tasks.py:290-292)
//
// Design (HBM-bound, no MFMA): every lane streams HS_V=4 consecutive rows per step with 16-byte loads
// (f32 x4, 2x i64 x2, u8 x4), consecutive lanes take consecutive row quads, so a wave reads 1 KiB
// per f32 column per instruction.  Group keys are resolved in a per-workgroup LDS dictionary; every
// lane owns a private accumulator table in LDS laid out [slot][acc][lane] (conflict-free ds_read_b64 /
// ds_write_b64, no atomics), reduced at the end in a fixed order -> bitwise reproducible results.
#include <stdlib.h>
#include <string.h>

#include "hs_agg_kernel.h"

// ahead-of-time instantiations: the bytecode interpreter (always available; also the reference point
// the JIT-compiled programs are tested against)
template <bool HASHED, int D>
__global__ void __launch_bounds__(256) k_agg_main(const AggMainArgs A_kernarg) {
    HS_KERNARG(AggMainArgs, A);
    hs_agg_main_body<InterpProg<HASHED, D>>(A);
}

// shared-dictionary tier (hs_agg_kernel.h): interpreter instantiations.  512 lanes, not the compiled programs' 1024: the
// interpreter's operand stack (D x 4 rows x 64 bits) does not fit 128 VGPRs - at 1024 lanes it lived in scratch memory
// (440-712 bytes per lane, ~5 400 scratch instructions).  A stack of depth 4 fits the 256 VGPRs of a 512-lane workgroup,
// depth 8 needs the accumulation registers as well: 256 lanes.  Chunks are whole steps of 1024 x 4 rows, so any of these
// widths walks them exactly.
static constexpr int hs_shared_interp_wg(int depth) { return depth > 4 ? 256 : 512; }
template <bool HASHED, int D>
__global__ void __launch_bounds__(hs_shared_interp_wg(D)) k_agg_shared(const AggMainArgs A_kernarg) {
    HS_KERNARG(AggMainArgs, A);
    hs_agg_shared_body<InterpProg<HASHED, D>>(A);
}

// unit tables of the shared tier: keys EMPTY, representative rows -1, cells = the aggregates' identities
struct SharedInitArgs {
    uint64_t* keys;
    int64_t* reps;
    uint64_t* acc;
    int64_t n_slots;  // n_units * unit_cap
    hs_agg_spec spec;
};
__global__ void __launch_bounds__(256) k_agg_shared_init(const SharedInitArgs A) {
    const int NA = A.spec.n_acc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.n_slots; i += (int64_t)gridDim.x * blockDim.x) {
        A.keys[i] = HS_EMPTY_KEY;
        A.reps[i] = -1;
        for (int a = 0; a < NA; ++a) A.acc[i * NA + a] = hs_acc_identity(A.spec.op[a], A.spec.is_int[a] != 0);
    }
}
// computed units: unit cell (unit, slot, a) = fold over the chunks' own cells.  One wave per (unit, slot): lane l folds
// chunks l, l + 64, ... in ascending order, then a fixed shuffle tree - the same order every run.
struct SharedFoldArgs {
    const uint64_t* chunk_acc;  // [n_chunks][n_units][unit_cap][n_acc]
    const int64_t* reps;        // [n_units][unit_cap]
    uint64_t* acc;              // [n_units][unit_cap][n_acc]
    int64_t n_chunks;
    int32_t n_units, unit_cap;
    hs_agg_spec spec;
};
__global__ void __launch_bounds__(256) k_agg_shared_fold_chunks(const SharedFoldArgs A) {
    // one wave per (unit, slot, aggregate): the aggregates of a cell are independent folds (round 2 walked them one
    // after the other in one wave: 4 x 14 dependent loads for config 4's 916 chunks)
    const int NA = A.spec.n_acc;
    const int lane = threadIdx.x & (HS_WAVE - 1);
    const int64_t task = (int64_t)blockIdx.x * (blockDim.x / HS_WAVE) + threadIdx.x / HS_WAVE;
    const int64_t cell = task / NA;  // (unit, slot)
    const int a = (int)(task - cell * NA);
    if (cell >= (int64_t)A.n_units * A.unit_cap || A.reps[cell] < 0) return;  // wave-uniform
    const int64_t stride = (int64_t)A.n_units * A.unit_cap * NA;
    const uint32_t op = A.spec.op[a];
    const bool is_int = A.spec.is_int[a] != 0;
    uint64_t v = hs_acc_identity(op, is_int);
    for (int64_t c = lane; c < A.n_chunks; c += HS_WAVE) v = hs_acc_fold(op, is_int, v, A.chunk_acc[c * stride + cell * NA + a]);
    for (int d = HS_WAVE / 2; d >= 1; d >>= 1) v = hs_acc_fold(op, is_int, v, hs_shfl_down64(v, d));
    if (lane == 0) A.acc[cell * NA + a] = v;
}

// after the scan: cells -> what the reference's shuffle file holds (f32 / i32 rounding, overflow and type flags),
// groups per unit.  One workgroup per unit.
struct SharedFinishArgs {
    const int64_t* reps;
    uint64_t* acc;
    int32_t* ngroups;
    int32_t unit_cap, pad;
    hs_agg_spec spec;
    uint32_t* flags;
};
__global__ void __launch_bounds__(256) k_agg_shared_finish(const SharedFinishArgs A) {
    __shared__ int s_count;
    const int NA = A.spec.n_acc, UC = A.unit_cap;
    const int64_t u = blockIdx.x;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    uint32_t err = 0;
    int mine = 0;
    for (int sl = threadIdx.x; sl < UC; sl += blockDim.x) {
        if (A.reps[u * UC + sl] < 0) continue;
        ++mine;
        for (int a = 0; a < NA; ++a) {
            uint64_t* cell = &A.acc[(u * UC + sl) * NA + a];
            const bool is_int = A.spec.is_int[a] != 0;
            if (hs_float_identity_left(A.spec.op[a], is_int, *cell)) err |= HS_FLAG_TYPE_ASSERT;
            *cell = hs_quantise_cell(is_int, *cell, err);
        }
    }
    if (mine) atomicAdd(&s_count, mine);
    __syncthreads();
    if (threadIdx.x == 0) A.ngroups[u] = s_count;
    if (err) atomicOr(A.flags, err);
}

// --------------------------------------------------------------------------------------------------
// One workgroup per unit: the stand-alone form of the unit combine (body: hs_agg_kernel.h).  Used when the combine is
// not fused into the scan kernel's epilogue (a unit without rows, tables too small to stage it).
__global__ void __launch_bounds__(256) k_agg_unit(const AggUnitArgs A_kernarg) {
    HS_KERNARG(AggUnitArgs, A);
    extern __shared__ __align__(16) uint64_t lds[];
    hs_agg_unit_body<false>(A, (int64_t)blockIdx.x, lds);
}

// --------------------------------------------------------------------------------------------------
struct AggPackArgs {
    const int64_t* rep;
    const uint64_t* acc;
    const int32_t* ngroups;
    int64_t n_units;
    int32_t group_cap;
    int32_t n_acc;
    int64_t* pack_start;  // [n_units+1]
    int64_t* out_rep;
    const int64_t* unit_ids;  // optional [n_units]: global id of every local unit (multi-GPU: file block id)
    int64_t* out_unit;        // optional: unit id of every packed row (orders the final merge across ranks)
    void* out_cols[HS_MAX_ACC];
    int32_t acc_kinds[HS_MAX_ACC];
};

// single workgroup: scan of per-unit group counts, then every thread packs whole units
__global__ void __launch_bounds__(256) k_agg_pack(const AggPackArgs A_kernarg) {
    HS_KERNARG(AggPackArgs, A);
    __shared__ int64_t s_scan[256];
    __shared__ int64_t s_base;
    const int tid = threadIdx.x;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int64_t u0 = 0; u0 < A.n_units; u0 += 256) {
        const int64_t u = u0 + tid;
        const int64_t n = u < A.n_units ? A.ngroups[u] : 0;
        s_scan[tid] = n;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {  // Hillis-Steele inclusive scan
            const int64_t v = tid >= d ? s_scan[tid - d] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        const int64_t start = s_base + s_scan[tid] - n;
        if (u < A.n_units) {
            A.pack_start[u] = start;
            int64_t o = start;
            for (int s = 0; s < A.group_cap; ++s) {
                const int64_t rep = A.rep[u * A.group_cap + s];
                if (rep < 0) continue;
                A.out_rep[o] = rep;
                if (A.out_unit) A.out_unit[o] = A.unit_ids ? A.unit_ids[u] : u;
                for (int a = 0; a < A.n_acc; ++a) {
                    const uint64_t cell = A.acc[(u * A.group_cap + s) * A.n_acc + a];
                    switch (A.acc_kinds[a]) {
                        case HS_F32: ((float*)A.out_cols[a])[o] = (float)hs_u2d(cell); break;
                        case HS_I32: ((int32_t*)A.out_cols[a])[o] = (int32_t)(int64_t)cell; break;
                        default: ((uint64_t*)A.out_cols[a])[o] = cell; break;
                    }
                }
                ++o;
            }
        }
        __syncthreads();
        if (tid == 255) s_base += s_scan[255];
        __syncthreads();
    }
    if (tid == 0) A.pack_start[A.n_units] = s_base;
}

// Wide units (hundreds of slots, shared tier): scan of the per-unit counts by one workgroup, then one workgroup per
// unit packs its occupied slots in slot order (ballot prefix per wave + a hop through LDS).
__global__ void __launch_bounds__(256) k_agg_pack_scan(const int32_t* ngroups, int64_t n_units, int64_t* pack_start) {
    __shared__ int64_t s_part[4];
    __shared__ int64_t s_base;
    const int tid = threadIdx.x, lane = tid & (HS_WAVE - 1), w = tid / HS_WAVE;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int64_t u0 = 0; u0 < n_units; u0 += 256) {
        const int64_t u = u0 + tid;
        const int64_t n = u < n_units ? ngroups[u] : 0;
        int64_t x = n;
        for (int d = 1; d < HS_WAVE; d <<= 1) {
            const int64_t t = __shfl_up(x, d, HS_WAVE);
            if (lane >= d) x += t;
        }
        if (lane == HS_WAVE - 1) s_part[w] = x;
        __syncthreads();
        int64_t before = 0, all = 0;
        for (int k = 0; k < 4; ++k) {
            if (k < w) before += s_part[k];
            all += s_part[k];
        }
        if (u < n_units) pack_start[u] = s_base + before + x - n;
        __syncthreads();
        if (tid == 0) s_base += all;
        __syncthreads();
    }
    if (tid == 0) pack_start[n_units] = s_base;
}
__global__ void __launch_bounds__(256) k_agg_pack_wide(const AggPackArgs A_kernarg) {
    HS_KERNARG(AggPackArgs, A);
    __shared__ int s_wave[4];
    __shared__ int s_run;
    const int tid = threadIdx.x, lane = tid & (HS_WAVE - 1), w = tid / HS_WAVE;
    const int64_t u = blockIdx.x;
    const int GC = A.group_cap;
    const int64_t out0 = A.pack_start[u];
    const int64_t uid = A.unit_ids ? A.unit_ids[u] : u;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int s0 = 0; s0 < GC; s0 += 256) {
        const int sl = s0 + tid;
        const int64_t rep = sl < GC ? A.rep[u * GC + sl] : -1;
        const unsigned long long m = __ballot(rep >= 0);
        if (lane == 0) s_wave[w] = __popcll(m);
        __syncthreads();
        int before = s_run;
        for (int k = 0; k < w; ++k) before += s_wave[k];
        if (rep >= 0) {
            const int64_t o = out0 + before + __popcll(m & ((1ull << lane) - 1));
            A.out_rep[o] = rep;
            if (A.out_unit) A.out_unit[o] = uid;
            for (int a = 0; a < A.n_acc; ++a) {
                const uint64_t cell = A.acc[(u * GC + sl) * A.n_acc + a];
                switch (A.acc_kinds[a]) {
                    case HS_F32: ((float*)A.out_cols[a])[o] = (float)hs_u2d(cell); break;
                    case HS_I32: ((int32_t*)A.out_cols[a])[o] = (int32_t)(int64_t)cell; break;
                    default: ((uint64_t*)A.out_cols[a])[o] = cell; break;
                }
            }
        }
        __syncthreads();
        if (tid == 0) s_run += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------------------
struct AggMergeArgs {
    hs_col key;
    hs_col acc_cols[HS_MAX_ACC];
    hs_agg_spec spec;
    int64_t n_rows;            // upper bound (sizes LDS)
    const int64_t* n_rows_dev; // optional exact row count on the device
    const int64_t* order;      // optional per-row order key (global unit id); rows with order < 0 are padding
    int32_t n_order;           // order keys lie in [0, n_order)
    int32_t limited;  // n_rows was cut down to what LDS holds: more rows on the device -> HS_FLAG_MERGE_ROWS
    int32_t cap;
    int32_t hashed;
    int64_t* out_rep;
    uint64_t* out_acc;
    int64_t* out_ngroups;
    uint32_t* flags;
};

// Inclusive scan of one int per thread over the workgroup (wave shuffles + one LDS hop): 2 barriers instead of
// the 2 * log2(n) of a Hillis-Steele ladder - the merge below is latency-bound.  s_part: >= 16 ints of LDS.
__device__ __forceinline__ int hs_block_scan_incl(int v, int* s_part, int& total) {
    const int tid = threadIdx.x, lane = tid & (HS_WAVE - 1), w = tid / HS_WAVE, nw = (blockDim.x + HS_WAVE - 1) / HS_WAVE;
    int x = v;
#pragma unroll
    for (int d = 1; d < HS_WAVE; d <<= 1) {
        const int t = __shfl_up(x, d, HS_WAVE);
        if (lane >= d) x += t;
    }
    if (lane == HS_WAVE - 1) s_part[w] = x;
    __syncthreads();
    int base = 0, all = 0;
    for (int k = 0; k < nw; ++k) {
        const int p = s_part[k];
        if (k < w) base += p;
        all += p;
    }
    __syncthreads();  // s_part may be reused
    total = all;
    return x + base;
}

// Row sources of the merge core.  Flat: the columns of a batch.  Slabs: `world` exchange slabs, row r =
// row r % M of slab r / M (hs_slab_desc).
struct MergeFlatIn {
    const AggMergeArgs& A;
    __device__ __forceinline__ bool has_order() const { return A.order != nullptr; }
    __device__ __forceinline__ int64_t order(int r) const { return A.order ? A.order[r] : 0; }
    __device__ __forceinline__ int upsert(uint64_t* dkeys, int64_t* dreps, uint32_t mask, int r) const {
        const uint64_t k = hs_key_at(A.key, r);
        return A.hashed ? hs_dict_upsert_rows(dreps, mask, A.key, k, r) : hs_dict_upsert_word(dkeys, dreps, mask, k, r);
    }
    __device__ __forceinline__ uint64_t cell(int a, int r) const { return hs_load_cell(A.acc_cols[a], r); }
    // handle of a row = what the sort-free merge keeps per (unit, group): here the row itself
    __device__ __forceinline__ int handle(int r) const { return r; }
    __device__ __forceinline__ int row_of(int h) const { return h; }
    __device__ __forceinline__ uint64_t cell_h(int a, int h) const { return cell(a, h); }
    // the stored kind of fold a's partials, decided once per fold (0: i32, 1: f32, 2: ask per cell), and the typed read
    __device__ __forceinline__ int kind_of(int a) const {
        const int k = A.acc_cols[a].kind;
        return k == HS_I32 ? 0 : k == HS_F32 ? 1 : 2;
    }
    __device__ __forceinline__ const void* cell_base(int a) const { return A.acc_cols[a].data; }
    template <int KIND>
    __device__ __forceinline__ uint64_t cell_hk(const void* base, int a, int h) const {
        if constexpr (KIND == 0) return (uint64_t)(int64_t)((const int32_t*)base)[h];
        else if constexpr (KIND == 1) return hs_d2u((double)((const float*)base)[h]);
        else return cell(a, h);
    }
};

struct AggFinishArgs {
    const uint8_t* slabs;
    hs_slab_desc desc;
    hs_finish_spec fin;
    hs_agg_spec spec;  // per fold: op / is_int
    hs_program prog;
    HsCols nocols;     // the projection reads merged cells through the sink; string operators never occur
    int32_t world, cap, n_order, key_bytes;
    uint8_t* result;
    int64_t* out_rep;     // scratch [cap]
    uint64_t* out_acc;    // scratch [n_fold][cap]
    uint8_t* key_scratch; // scratch [cap][key_bytes]
    uint32_t* flags;
    uint32_t* own_slab_flags;
    // small inputs keep their whole working set in LDS (every global hop of this one-workgroup kernel is a
    // serialised ~2 us round trip): byte offsets into the dynamic LDS block, 0 = stay in global memory
    int64_t lds_slabs, lds_work, lds_image, lds_stack, image_bytes;
    int64_t* stamps;  // optional [16]: wall_clock64() at the phase boundaries (HIPSPARK_FINISH_STAMPS=1; tools/finish_phases.py)
    // The usual projection after a merge is a handful of "a <op> b" over two merged columns (AVG = sum / count):
    // such a program is decoded on the host into these entries and evaluated by (group, entry) lanes directly -
    // the interpreter's ~6 us (instruction-cache misses of a kernel that runs once) shrink to well under one.
    int32_t n_simple, pad_simple;
    struct {
        int16_t out, op, a, b;  // outs[] index, HS_OP_*_F / *_I opcode, program column slots
        int8_t conv_a, conv_b, pad0, pad1;  // int -> float conversion of the operand (HS_OP_I2F)
    } simple[HS_MAX_OUTS];
};
// where a launch keeps its working set
struct FinishView {
    const uint8_t* slabs;   // world slabs, desc.stride apart
    int64_t* out_rep;       // [cap]
    uint64_t* out_acc;      // [n_fold][cap]
    uint8_t* key_scratch;   // [cap][key_bytes]
    uint8_t* image;         // the result image (or its LDS staging copy)
};
struct MergeSlabIn {
    const AggFinishArgs& A;
    const FinishView& V;
    __device__ __forceinline__ const uint8_t* slab_of(int r, int& i) const {
        const int k = r / (int)A.desc.slab_rows;
        i = r - k * (int)A.desc.slab_rows;
        return V.slabs + (int64_t)k * A.desc.stride;
    }
    __device__ __forceinline__ bool has_order() const { return true; }
    __device__ __forceinline__ int64_t order(int r) const {
        int i;
        const uint8_t* s = slab_of(r, i);
        return ((const int64_t*)(s + A.desc.order_off))[i];
    }
    __device__ __forceinline__ hs_col key_col(const uint8_t* s) const {
        return hs_col{A.desc.key_kind, A.desc.key_kind == HS_STR ? A.desc.key_len : -1, s + A.desc.key_off, nullptr, nullptr};
    }
    __device__ __forceinline__ int upsert(uint64_t* dkeys, int64_t* dreps, uint32_t mask, int r) const {
        int i;
        const uint8_t* s = slab_of(r, i);
        return hs_dict_upsert_word(dkeys, dreps, mask, hs_key_at(key_col(s), i), r);
    }
    __device__ __forceinline__ uint64_t cell(int a, int r) const {
        int i;
        const uint8_t* s = slab_of(r, i);
        const int src = A.fin.fold_src[a];
        const uint8_t* base = s + A.desc.acc_off[src];
        if (A.desc.acc_kind[src] == HS_I32) return (uint64_t)(int64_t)((const int32_t*)base)[i];
        return hs_d2u((double)((const float*)base)[i]);
    }
    // handle = the row's element index relative to ANY 4-byte column of slab 0 (slabs are stride bytes apart, stride a
    // multiple of 16): the fold reads cells without dividing by the slab length again
    __device__ __forceinline__ int handle(int r) const {
        const int k = r / (int)A.desc.slab_rows;
        return k * (int)(A.desc.stride >> 2) + (r - k * (int)A.desc.slab_rows);
    }
    __device__ __forceinline__ int row_of(int h) const {
        const int k = h / (int)(A.desc.stride >> 2);
        return k * (int)A.desc.slab_rows + (h - k * (int)(A.desc.stride >> 2));
    }
    __device__ __forceinline__ uint64_t cell_h(int a, int h) const {
        const int src = A.fin.fold_src[a];
        const uint8_t* base = V.slabs + A.desc.acc_off[src];
        if (A.desc.acc_kind[src] == HS_I32) return (uint64_t)(int64_t)((const int32_t*)base)[h];
        return hs_d2u((double)((const float*)base)[h]);
    }
    __device__ __forceinline__ int kind_of(int a) const { return A.desc.acc_kind[A.fin.fold_src[a]] == HS_I32 ? 0 : 1; }
    __device__ __forceinline__ const void* cell_base(int a) const { return V.slabs + A.desc.acc_off[A.fin.fold_src[a]]; }
    template <int KIND>
    __device__ __forceinline__ uint64_t cell_hk(const void* base, int, int h) const {
        if constexpr (KIND == 0) return (uint64_t)(int64_t)((const int32_t*)base)[h];
        else return hs_d2u((double)((const float*)base)[h]);
    }
};

// Ordered fold of one group's bucket (cells q = b .. e-1 of aggregate a), specialised on the aggregate so that
// the loop body is the bare dependent chain (one fp64 add per cell for SUM): the fold runs on a handful of
// lanes of one wave, where every extra instruction is serialised latency.  Four LDS reads in flight.
template <uint32_t OP, bool IS_INT>
__device__ __forceinline__ uint64_t hs_fold_bucket(const uint64_t* sorted, int b, int e, int NA, int a) {
    constexpr int W = 8;  // cells per batch; the next batch is in flight while this one is folded (LDS latency
                          // ~130 cycles vs ~8 per dependent fp64 add)
    uint64_t v = hs_acc_identity(OP, IS_INT);
    const uint64_t* p = sorted + (int64_t)b * NA + a;
    int left = e - b;
    if (left >= W) {
        uint64_t c[W], nx[W];
#pragma unroll
        for (int k = 0; k < W; ++k) c[k] = p[k * NA];
        p += W * NA;
        left -= W;
        while (left >= W) {
#pragma unroll
            for (int k = 0; k < W; ++k) nx[k] = p[k * NA];
            p += W * NA;
            left -= W;
#pragma unroll
            for (int k = 0; k < W; ++k) v = hs_acc_fold(OP, IS_INT, v, c[k]);
#pragma unroll
            for (int k = 0; k < W; ++k) c[k] = nx[k];
        }
#pragma unroll
        for (int k = 0; k < W; ++k) v = hs_acc_fold(OP, IS_INT, v, c[k]);
    }
    for (; left > 0; --left, p += NA) v = hs_acc_fold(OP, IS_INT, v, *p);
    return v;
}

// ---- wave-wide ordered fold (long buckets) ----------------------------------------------------------------------
// The reference folds a group's partials front to back (fp64, aggregate.py:71-84); one lane doing that for a bucket
// of a few hundred cells is a chain of dependent adds that nothing else in the launch can hide.  A whole wave takes
// the bucket instead, without giving up the reference's result:
//   MIN / MAX, integer SUM  combine(left, right) with the earlier part on the left is associative (ties keep the
//                           left operand, integer adds wrap) - a tree in lane order IS the sequential fold;
//   float SUM               every addition carries its rounding error (Knuth TwoSum).  If the lane-chunk sums, the
//                           scan over the lanes AND every prefix 0 + x0 + ... + xk rebuilt from them are all exact,
//                           each prefix is a representable number and the sequential chain produces exactly those -
//                           the total is the reference's bit for bit.  Any inexact step (or inf / nan) -> `exact`
//                           is false and the caller folds sequentially.  Partials are f32 values of similar
//                           magnitude, so the exact case is the usual one.
__device__ __forceinline__ bool hs_add_exact(double a, double b, double& s) {
    s = a + b;
    const double bb = s - a;
    const double err = (a - (s - bb)) + (b - bb);
    return err == 0.0;
}

// -> the folded cell (valid on every lane when `exact`); b = first bucket position, m = bucket length
__device__ __forceinline__ uint64_t hs_fold_bucket_wave(uint32_t op, bool is_int, const uint64_t* sorted, int b, int m, int NA,
                                                        int a, int lane, bool& exact) {
    const int c = (m + HS_WAVE - 1) / HS_WAVE;
    const int lo = lane * c < m ? lane * c : m, hi = (lo + c) < m ? (lo + c) : m;
    const uint64_t* p = sorted + (int64_t)(b + lo) * NA + a;
    exact = true;
    if (is_int || op != HS_AGG_SUM) {
        uint64_t v = hs_acc_identity(op, is_int);
        for (int j = 0; j < hi - lo; ++j) v = hs_acc_fold(op, is_int, v, p[(int64_t)j * NA]);
        for (int d = 1; d < HS_WAVE; d <<= 1) {  // lane L holds lanes L .. L+2d-1 after the step; lane 0 ends with all
            const uint64_t other = __shfl_down(v, d, HS_WAVE);
            if (lane + d < HS_WAVE) v = hs_acc_fold(op, is_int, v, other);
        }
        return __shfl(v, 0, HS_WAVE);
    }
    bool ok = true;
    double s = 0.0;
    for (int j = 0; j < hi - lo; ++j) ok &= hs_add_exact(s, hs_u2d(p[(int64_t)j * NA]), s);
    double incl = s;
    for (int d = 1; d < HS_WAVE; d <<= 1) {
        const double before = __shfl_up(incl, d, HS_WAVE);
        if (lane >= d) ok &= hs_add_exact(before, incl, incl);
    }
    double q = __shfl_up(incl, 1, HS_WAVE);
    if (lane == 0) q = 0.0;
    for (int j = 0; j < hi - lo; ++j) ok &= hs_add_exact(q, hs_u2d(p[(int64_t)j * NA]), q);
    exact = __all(ok);
    return hs_d2u(__shfl(incl, HS_WAVE - 1, HS_WAVE));
}

// One (group, aggregate) of the sort-free merge by ONE lane: the reference's own chain 0 + p(unit 0) + p(unit 1) + ... in unit
// order (tasks.py:290-292), so there is nothing to verify - the cost is the chain's latency.  The aggregate kind is a template
// argument and eight units' row handles, then their cells, are read together, which leaves one dependent fold per unit
// per unit.  Measured (profiles/r04_finish_kernel_phases.txt, fold phase): 3 units 2.0 us, 36 units 6.4 us, 287 units 11.5 us -
// against 4.6 / 7.0-7.4 / 7.7 us for the lane-group protocol below, whose price is its instruction count (a cold
// one-workgroup launch pays per instruction fetched), not its length: the chain wins for few units only.
// A unit without the key contributes the identity (see hs_fold_group_indexed).
template <uint32_t OP, bool IS_INT, int KIND, class In>
__device__ __forceinline__ uint64_t hs_fold_units_seq_k(const In& in, const int32_t* M, int cap, int sl, int nord, int a) {
    const uint64_t ident = hs_acc_identity(OP, IS_INT);
    const void* base = in.cell_base(a);  // (read once: the argument block lives in LDS, every field is a round trip)
    const int32_t* Ms = M + sl;
    uint64_t v = ident;
    for (int o0 = 0; o0 < nord; o0 += 8) {
        int h[8];
        uint64_t x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int o = o0 + k < nord ? o0 + k : nord - 1;
            const int got = Ms[o * cap];
            h[k] = o0 + k < nord ? got : -1;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = in.template cell_hk<KIND>(base, a, h[k] < 0 ? 0 : h[k]);  // (handle 0 = row 0 of slab 0: always there)
#pragma unroll
        for (int k = 0; k < 8; ++k) v = hs_acc_fold(OP, IS_INT, v, h[k] >= 0 ? x[k] : ident);
    }
    return v;
}
// (the partials' stored kind picks the loop once: a per-cell kind test kept the eight reads from travelling together)
template <uint32_t OP, bool IS_INT, class In>
__device__ __forceinline__ uint64_t hs_fold_units_seq(const In& in, const int32_t* M, int cap, int sl, int nord, int a, int kind) {
    if (kind == 0) return hs_fold_units_seq_k<OP, IS_INT, 0>(in, M, cap, sl, nord, a);
    if (kind == 1) return hs_fold_units_seq_k<OP, IS_INT, 1>(in, M, cap, sl, nord, a);
    return hs_fold_units_seq_k<OP, IS_INT, 2>(in, M, cap, sl, nord, a);
}
constexpr int HS_SEQ_FOLD_MAX = 64;  // units up to which a lane's chain beats the lane-group protocol

// The same fold over a sequence given by a functor, by a GROUP of `width` lanes (16, 32 or 64: several folds share a
// wave when there are more (group, aggregate) pairs than waves, or few positions per fold).  get(i) = cell i of the
// sequence, or the aggregate's identity for a position that holds nothing (SUM: adding +0.0 / 0 changes neither the
// value - the accumulator starts from the identity and so is never -0.0 - nor the exactness bookkeeping; MIN / MAX:
// folding the identity in again is a no-op, the reference's accumulators start from it, tasks.py:303-310).  Every lane
// of the wave must call this (m = 0 for a group without work); the result is valid on all lanes of the group.
template <class Get>
__device__ __forceinline__ uint64_t hs_fold_group_indexed(uint32_t op, bool is_int, const Get& get, int m, int lane, int width,
                                                          bool& exact) {
    const int ln = lane & (width - 1), first_lane = lane - ln;
    const int c = (m + width - 1) / width;
    const int lo = ln * c < m ? ln * c : m, hi = (lo + c) < m ? (lo + c) : m;
    exact = true;
    if (is_int || op != HS_AGG_SUM) {
        uint64_t v = hs_acc_identity(op, is_int);
        for (int i = lo; i < hi; ++i) v = hs_acc_fold(op, is_int, v, get(i));
        for (int d = 1; d < width; d <<= 1) {  // lane L holds lanes L .. L+2d-1 after the step; the group's lane 0 ends with all
            const uint64_t other = __shfl_down(v, d, width);
            if (ln + d < width) v = hs_acc_fold(op, is_int, v, other);
        }
        return __shfl(v, first_lane, HS_WAVE);
    }
    bool ok = true;
    double s = 0.0;
    constexpr int KEEP = 12;  // a lane's cells stay in registers for the second pass
    double kept[KEEP];
    if (c <= KEEP) {
#pragma unroll
        for (int j = 0; j < KEEP; ++j) kept[j] = lo + j < hi ? hs_u2d(get(lo + j)) : 0.0;
#pragma unroll
        for (int j = 0; j < KEEP; ++j) ok &= hs_add_exact(s, kept[j], s);
    } else {
        for (int i = lo; i < hi; ++i) ok &= hs_add_exact(s, hs_u2d(get(i)), s);
    }
    double incl = s;
    for (int d = 1; d < width; d <<= 1) {
        const double before = __shfl_up(incl, d, width);
        if (ln >= d) ok &= hs_add_exact(before, incl, incl);
    }
    double q = __shfl_up(incl, 1, width);
    if (ln == 0) q = 0.0;
    if (c <= KEEP) {
#pragma unroll
        for (int j = 0; j < KEEP; ++j) ok &= hs_add_exact(q, kept[j], q);
    } else {
        for (int i = lo; i < hi; ++i) ok &= hs_add_exact(q, hs_u2d(get(i)), q);
    }
    const unsigned long long mine = width == HS_WAVE ? ~0ull : (((1ull << width) - 1) << first_lane);
    exact = (__ballot(ok) & mine) == mine;
    return hs_d2u(__shfl(incl, first_lane + width - 1, HS_WAVE));
}

// Final merge, everything staged in LDS.  The partials of a group must be folded in the reference's
// order: ascending (order key, row) - with no order keys simply ascending row - i.e. block order of the
// shuffle file (0 + p_block0 + p_block1 + ... in fp64).  All steps are O(n):
//   1. rows -> dictionary slot (parallel);
//   2. visiting sequence = rows sorted by (order key, row): a counting sort on the order key (block id);
//      rows of one block are contiguous in the input, so the position inside a block is r - first_row;
//   3. one wave walks the sequence 64 rows at a time and ranks the rows of each group with a
//      ballot / match-any loop (stable);
//   4. cells are scattered to [group bucket][rank], then one lane per (group, aggregate) folds its
//      bucket front to back from contiguous LDS.
// Outputs are dense (groups in ascending dictionary-slot order) and column-major: out_acc[a * cap + i].
// One workgroup; returns the number of groups (uniform).
template <typename In>
__device__ __forceinline__ int hs_merge_small_core(const In& in, const hs_agg_spec& spec, const int n, const int64_t nmax,
                                                   const int nord, const int cap, uint64_t* lds, int64_t* out_rep,
                                                   uint64_t* out_acc, uint32_t& err, int64_t* stamps = nullptr) {
#define HS_STAMP(i) do { if (stamps && threadIdx.x == 0) stamps[i] = (int64_t)wall_clock64(); } while (0)
    const int NA = spec.n_acc;
    uint64_t* dkeys = lds;                                // [cap]
    int64_t* dreps = (int64_t*)(lds + cap);               // [cap]
    uint64_t* sorted = lds + 2 * cap;                     // [nmax][NA] cells bucketed by group
    int32_t* rslot = (int32_t*)(sorted + nmax * NA);      // [nmax] row -> slot (-1: padding / overflow)
    int32_t* seq = rslot + nmax;                          // [nmax] visiting sequence: seq[i] = row
    int32_t* rrank = seq + nmax;                          // [nmax] row -> rank inside its group
    int32_t* dense = rrank + nmax;                        // [cap] slot -> dense output row
    int32_t* cnt = dense + cap;                           // [cap] rows per slot
    int32_t* start = cnt + cap;                           // [cap] first bucket position of the slot
    int32_t* run = start + cap;                           // [cap] running rank per slot (step 3)
    int32_t* ocnt = run + cap;                            // [nord] rows per order key
    int32_t* ofirst = ocnt + nord;                        // [nord] first row of an order key
    __shared__ int s_part[16];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const bool ordered = in.has_order();

    for (int i = tid; i < cap; i += nthr) {
        dkeys[i] = HS_EMPTY_KEY;
        dreps[i] = -1;
        cnt[i] = 0;
        run[i] = 0;
        out_rep[i] = -1;  // a group that never gets its first row shows up as an invalid row index downstream
                          // (checked gathers raise HS_FLAG_BAD_PROGRAM) instead of as stale memory
    }
    for (int i = tid; i < nord; i += nthr) {
        ocnt[i] = 0;
        ofirst[i] = 0x7fffffff;
    }
    __syncthreads();
    const uint32_t mask = (uint32_t)cap - 1;
    // ---- round 3: the ordered merge without a sort ----------------------------------------------------------------
    // Partial rows carry their unit (order key), and a unit holds a key at most once: the partials of group g in merge
    // order are simply M[o][g] for o = 0 .. nord-1 (a row or nothing).  One scatter builds M; every (group, aggregate)
    // is then folded by a wave (or a lane, when there are few units) straight from the staged rows.  The counting
    // sort + visiting sequence + ranking + bucket scatter of the general form (below; still used when M does not fit
    // the bucket area, or when two rows of one unit do share a key) were 10 of the finish launch's 23 us at sf=100.
    if (ordered && (int64_t)nord * cap * 4 <= nmax * ((int64_t)NA * 8 + 12)) {
        __shared__ int s_dup;
        int32_t* M = (int32_t*)sorted;  // [nord][cap], aliases the bucket area (+ rslot / seq / rrank)
        for (int i = tid; i < nord * cap; i += nthr) M[i] = -1;
        if (tid == 0) s_dup = 0;
        __syncthreads();
        uint32_t ferr = 0;
        for (int r = tid; r < n; r += nthr) {
            const int64_t o = in.order(r);
            if (o < 0) continue;
            const int sl = in.upsert(dkeys, dreps, mask, r);
            if (sl < 0) {
                ferr |= HS_FLAG_MERGE_FULL;
            } else if (o >= nord) {
                ferr |= HS_FLAG_BAD_PROGRAM;
            } else {
                atomicAdd(&cnt[sl], 1);
                if (atomicCAS(&M[(int)o * cap + sl], -1, in.handle(r)) != -1) s_dup = 1;
            }
        }
        __syncthreads();
        HS_STAMP(2);
        if (!s_dup) {
            err |= ferr;
            int ngroups;
            if (cap <= HS_WAVE) {  // the usual few groups: one ballot numbers them
                if (tid < HS_WAVE) {
                    const bool has = tid < cap && cnt[tid] > 0;
                    const unsigned long long m = __ballot(has);
                    if (has) run[__popcll(m & ((1ull << tid) - 1))] = tid;  // dense output row -> slot
                    if (tid == 0) s_part[0] = __popcll(m);
                }
                __syncthreads();
                ngroups = s_part[0];
                __syncthreads();  // (s_part is reused by later scans)
            } else {
                const int per = (cap + nthr - 1) / nthr;
                const int s0 = tid * per, s1 = (s0 + per) < cap ? (s0 + per) : cap;
                int mine = 0;
                for (int sl = s0; sl < s1; ++sl) mine += cnt[sl] > 0;
                int drun = hs_block_scan_incl(mine, s_part, ngroups) - mine;
                for (int sl = s0; sl < s1; ++sl) {
                    if (cnt[sl] > 0) run[drun++] = sl;
                }
                __syncthreads();
            }
            HS_STAMP(3);
            HS_STAMP(4);
            HS_STAMP(5);
            HS_STAMP(6);
            const int lane = tid & (HS_WAVE - 1), wv = tid / HS_WAVE, nwv = nthr / HS_WAVE;
            if (nord <= HS_SEQ_FOLD_MAX) {
                // round 4: a wave per aggregate (its kind is then wave-uniform: scalar dispatch, no divergence), a lane per
                // group, the reference's chain; one more wave finds every group's first row (the key's representative)
                for (int a = wv; a <= NA; a += nwv) {
                    if (a == NA) {
                        for (int g = lane; g < ngroups; g += HS_WAVE) {
                            const int sl = run[g];
                            int first = -1;
                            for (int o = 0; o < nord && first < 0; ++o) first = M[o * cap + sl];
                            out_rep[g] = first >= 0 ? in.row_of(first) : -1;
                        }
                        continue;
                    }
                    const uint32_t op = (uint32_t)__builtin_amdgcn_readfirstlane((int)spec.op[a]);
                    const bool is_int = __builtin_amdgcn_readfirstlane((int)spec.is_int[a]) != 0;
                    const int kind = __builtin_amdgcn_readfirstlane(in.kind_of(a));
                    for (int g = lane; g < ngroups; g += HS_WAVE) {
                        const int sl = run[g];
                        uint64_t v;
                        if (is_int) {
                            if (op == HS_AGG_SUM) v = hs_fold_units_seq<HS_AGG_SUM, true>(in, M, cap, sl, nord, a, kind);
                            else if (op == HS_AGG_MIN) v = hs_fold_units_seq<HS_AGG_MIN, true>(in, M, cap, sl, nord, a, kind);
                            else v = hs_fold_units_seq<HS_AGG_MAX, true>(in, M, cap, sl, nord, a, kind);
                        } else {
                            if (op == HS_AGG_SUM) v = hs_fold_units_seq<HS_AGG_SUM, false>(in, M, cap, sl, nord, a, kind);
                            else if (op == HS_AGG_MIN) v = hs_fold_units_seq<HS_AGG_MIN, false>(in, M, cap, sl, nord, a, kind);
                            else v = hs_fold_units_seq<HS_AGG_MAX, false>(in, M, cap, sl, nord, a, kind);
                        }
                        out_acc[(int64_t)a * cap + g] = v;
                    }
                }
                HS_STAMP(7);
                return ngroups;
            }
            // One fold per (group, aggregate), by a lane group of 16 / 32 / 64 lanes: as wide as possible while all folds
            // still fit one round of the workgroup's waves, and no wider than the sequence is long.  A further task per
            // group finds its first row (the representative the key is copied from).
            const int per_group = NA + 1, ntasks = ngroups * per_group;
            int width = HS_WAVE;
            while (width > 16 && (ntasks > nwv * (HS_WAVE / width) || nord <= width / 2)) width >>= 1;
            const int per_wave = HS_WAVE / width, nslots = nwv * per_wave, ln = lane & (width - 1);
            for (int c0 = 0; c0 < ntasks; c0 += nslots) {  // uniform trip count: every lane reaches the shuffles
                const int c = c0 + wv * per_wave + lane / width;
                const bool valid = c < ntasks;
                const int g = valid ? c / per_group : 0, a = valid ? c - g * per_group - 1 : 0;  // a == -1: the first row
                const int sl = valid ? run[g] : 0;
                if (a < 0) {
                    int first = 0x7fffffff;
                    if (valid)
                        for (int o = ln; o < nord; o += width)
                            if (M[o * cap + sl] >= 0) { first = o; break; }
                    for (int d = width / 2; d >= 1; d >>= 1) {
                        const int other = __shfl_down(first, d, width);
                        first = other < first ? other : first;
                    }
                    if (valid && ln == 0) out_rep[g] = first < nord ? in.row_of(M[first * cap + sl]) : -1;
                }
                // (groups of a wave may differ in `a`: the fold below is executed by all of them, a < 0 folds nothing)
                const int aa = a < 0 ? 0 : a;
                const uint32_t op = NA > 0 ? spec.op[aa] : HS_AGG_SUM;
                const bool is_int = NA > 0 ? spec.is_int[aa] != 0 : true;
                const uint64_t ident = hs_acc_identity(op, is_int);
                auto get = [&](int o) -> uint64_t {
                    const int h = M[o * cap + sl];
                    return h >= 0 ? in.cell_h(aa, h) : ident;
                };
                bool exact;
                uint64_t v = hs_fold_group_indexed(op, is_int, get, (valid && a >= 0) ? nord : 0, lane, width, exact);
                if (valid && a >= 0 && ln == 0) {
                    if (!exact) {  // the reference's own chain (fp64 additions in unit order)
                        double acc = 0.0;
                        for (int o = 0; o < nord; ++o) {
                            const int h = M[o * cap + sl];
                            if (h >= 0) acc = acc + hs_u2d(in.cell_h(a, h));
                        }
                        v = hs_d2u(acc);
                    }
                    out_acc[(int64_t)a * cap + g] = v;
                }
            }
            HS_STAMP(7);
            return ngroups;
        }
        // two rows of one unit share a key (not a well-formed shuffle file; the general form does not mind): start over
        __syncthreads();
        for (int i = tid; i < cap; i += nthr) {
            dkeys[i] = HS_EMPTY_KEY;
            dreps[i] = -1;
            cnt[i] = 0;
        }
        __syncthreads();
    }
    for (int r = tid; r < n; r += nthr) {  // step 1
        int sl = -1;
        const int64_t o = in.order(r);
        if (o >= 0) {
            sl = in.upsert(dkeys, dreps, mask, r);
            if (sl < 0) err |= HS_FLAG_MERGE_FULL;  // the merge's own capacity, not the per-unit dictionaries'
            else atomicAdd(&cnt[sl], 1);
            if (ordered && sl >= 0) {
                if (o >= nord) {
                    err |= HS_FLAG_BAD_PROGRAM;
                    sl = -1;
                } else {
                    atomicAdd(&ocnt[(int)o], 1);
                    atomicMin(&ofirst[(int)o], r);
                }
            }
        }
        rslot[r] = sl;
    }
    __syncthreads();
    HS_STAMP(2);
    // dense group numbering + bucket starts: per-thread slot ranges and two block scans
    const int per = (cap + nthr - 1) / nthr;
    const int s0 = tid * per, s1 = (s0 + per) < cap ? (s0 + per) : cap;
    int mine = 0, mine_rows = 0;
    for (int sl = s0; sl < s1; ++sl) {
        mine += cnt[sl] > 0;
        mine_rows += cnt[sl];
    }
    int ngroups, total_rows;
    int drun = hs_block_scan_incl(mine, s_part, ngroups) - mine;
    for (int sl = s0; sl < s1; ++sl) dense[sl] = cnt[sl] > 0 ? drun++ : -1;
    int pos = hs_block_scan_incl(mine_rows, s_part, total_rows) - mine_rows;
    for (int sl = s0; sl < s1; ++sl) {
        start[sl] = pos;
        pos += cnt[sl];
    }
    __syncthreads();

    HS_STAMP(3);
    // step 2: visiting sequence
    int nseq = n;
    if (ordered) {
        const int operth = (nord + nthr - 1) / nthr;
        const int o0 = tid * operth, o1 = (o0 + operth) < nord ? (o0 + operth) : nord;
        int osum = 0;
        for (int o = o0; o < o1; ++o) osum += ocnt[o];
        int opos = hs_block_scan_incl(osum, s_part, nseq) - osum;
        for (int o = o0; o < o1; ++o) {  // ocnt becomes the start of the block's run in the sequence
            const int c = ocnt[o];
            ocnt[o] = opos;
            opos += c;
        }
        __syncthreads();
        // Position of a row inside its block = number of VALID rows of the block before it.  The rows of a block
        // are contiguous in the input but not all valid (a row whose key did not fit a full dictionary, an unused
        // slab row): r - first_row would leave holes and push later rows out of the block's range.  Exclusive
        // count of valid rows before every row (kept in rrank[], which step 3 overwrites): thread ranges + one scan.
        const int rper = (n + nthr - 1) / nthr;
        const int r0 = tid * rper < n ? tid * rper : n, r1 = (r0 + rper) < n ? (r0 + rper) : n;
        int vmine = 0;
        for (int r = r0; r < r1; ++r) vmine += rslot[r] >= 0;
        int vtotal;
        int vrun = hs_block_scan_incl(vmine, s_part, vtotal) - vmine;
        for (int r = r0; r < r1; ++r) {
            rrank[r] = vrun;
            vrun += rslot[r] >= 0;
        }
        __syncthreads();
        for (int r = tid; r < n; r += nthr) {
            if (rslot[r] < 0) continue;
            const int o = (int)in.order(r);
            seq[ocnt[o] + (rrank[r] - rrank[ofirst[o]])] = r;
        }
        __syncthreads();
    }

    HS_STAMP(4);
    // step 3: stable rank of every row inside its group.  The sequence is cut into 64-row segments; a wave ranks
    // the rows of a segment with a ballot / match-any loop and notes the segment's count per group, a scan over
    // the segments of each group turns the counts into bases.  (The bucket area is free until step 4; when the
    // segment table does not fit there, one wave walks the whole sequence instead.)
    const int nseg = (nseq + HS_WAVE - 1) / HS_WAVE;
    const int lane = tid & (HS_WAVE - 1), wv = tid / HS_WAVE, nwv = nthr / HS_WAVE;
    if ((int64_t)nseg * cap <= nmax * NA * 2) {
        int32_t* segcnt = (int32_t*)sorted;  // [nseg][cap]
        for (int i = tid; i < nseg * cap; i += nthr) segcnt[i] = 0;
        __syncthreads();
        for (int seg = wv; seg < nseg; seg += nwv) {
            const int i = seg * HS_WAVE + lane;
            const int r = i < nseq ? (ordered ? seq[i] : i) : -1;
            const int sl = r >= 0 ? rslot[r] : -1;
            unsigned long long todo = __ballot(sl >= 0);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int g = __shfl(sl, leader, HS_WAVE);
                const unsigned long long same = __ballot(sl == g);
                if (sl == g) rrank[r] = __popcll(same & ((1ull << lane) - 1));
                if (lane == leader) segcnt[seg * cap + g] = __popcll(same);
                todo &= ~same;
            }
        }
        __syncthreads();
        for (int sl = tid; sl < cap; sl += nthr) {
            int running = 0;
            for (int seg = 0; seg < nseg; ++seg) {
                const int c = segcnt[seg * cap + sl];
                segcnt[seg * cap + sl] = running;
                running += c;
            }
        }
        __syncthreads();
        for (int i = tid; i < nseq; i += nthr) {
            const int r = ordered ? seq[i] : i;
            const int sl = rslot[r];
            if (sl >= 0) rrank[r] += segcnt[(i / HS_WAVE) * cap + sl];
        }
    } else if (tid < HS_WAVE) {
        for (int base = 0; base < nseq; base += HS_WAVE) {
            const int i = base + tid;
            const int r = i < nseq ? (ordered ? seq[i] : i) : -1;
            const int sl = r >= 0 ? rslot[r] : -1;
            unsigned long long todo = __ballot(sl >= 0);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int g = __shfl(sl, leader, HS_WAVE);
                const unsigned long long same = __ballot(sl == g);
                if (sl == g) rrank[r] = run[g] + __popcll(same & ((1ull << tid) - 1));
                if (tid == leader) run[g] += __popcll(same);
                todo &= ~same;
            }
        }
    }
    __syncthreads();
    HS_STAMP(5);
    // step 4: scatter cells into the group buckets, then the ordered fold
    for (int r = tid; r < n; r += nthr) {
        const int sl = rslot[r];
        if (sl < 0) continue;
        const int rk = rrank[r];
        const int at = start[sl] + rk;
        for (int a = 0; a < NA; ++a) sorted[(int64_t)at * NA + a] = in.cell(a, r);
        if (rk == 0) out_rep[dense[sl]] = r;  // first row of the group in merge order: race-independent
    }
    for (int sl = tid; sl < cap; sl += nthr)  // dense output row -> slot (run[] is free after step 3)
        if (dense[sl] >= 0) run[dense[sl]] = sl;
    __syncthreads();
    HS_STAMP(6);
    // Long buckets (>= 96 partials per group on average: few groups, many units) are folded by a wave each,
    // short ones by a lane each (measured on Q1: 36 partials per group 4.6 us by lanes / 5.9 us by waves,
    // 287 partials 8.3 us / 3.7 us).
    const bool wide = ngroups > 0 && (int64_t)ngroups * 96 <= (int64_t)total_rows;
    if (wide) {
        for (int c = wv; c < ngroups * NA; c += nwv) {
            const int g = c / NA, a = c - g * NA;
            const int sl = run[g];
            const uint32_t op = spec.op[a];
            const bool is_int = spec.is_int[a] != 0;
            const int b = start[sl], m = cnt[sl];
            bool exact;
            uint64_t v = hs_fold_bucket_wave(op, is_int, sorted, b, m, NA, a, lane, exact);
            if (!exact && lane == 0) v = hs_fold_bucket<HS_AGG_SUM, false>(sorted, b, b + m, NA, a);
            if (lane == 0) out_acc[(int64_t)a * cap + g] = v;
        }
    } else {
        for (int i = tid; i < cap * NA; i += nthr) {
            const int sl = i / NA, a = i % NA;
            if (dense[sl] < 0) continue;
            const uint32_t op = spec.op[a];
            const bool is_int = spec.is_int[a] != 0;
            const int b = start[sl], e = b + cnt[sl];
            uint64_t v;
            if (op == HS_AGG_SUM) v = is_int ? hs_fold_bucket<HS_AGG_SUM, true>(sorted, b, e, NA, a) : hs_fold_bucket<HS_AGG_SUM, false>(sorted, b, e, NA, a);
            else if (op == HS_AGG_MIN) v = is_int ? hs_fold_bucket<HS_AGG_MIN, true>(sorted, b, e, NA, a) : hs_fold_bucket<HS_AGG_MIN, false>(sorted, b, e, NA, a);
            else v = is_int ? hs_fold_bucket<HS_AGG_MAX, true>(sorted, b, e, NA, a) : hs_fold_bucket<HS_AGG_MAX, false>(sorted, b, e, NA, a);
            out_acc[(int64_t)a * cap + dense[sl]] = v;
        }
    }
    HS_STAMP(7);
    return ngroups;
#undef HS_STAMP
}

__global__ void __launch_bounds__(1024) k_agg_merge_small(const AggMergeArgs A_kernarg) {
    HS_KERNARG(AggMergeArgs, A);
    extern __shared__ __align__(16) uint64_t lds[];
    int64_t n64 = A.n_rows;
    uint32_t err = 0;
    if (A.n_rows_dev) {
        const int64_t nd = *A.n_rows_dev;
        n64 = nd < A.n_rows ? nd : A.n_rows;
        if (A.limited && nd > A.n_rows) err |= HS_FLAG_MERGE_ROWS;
    }
    const MergeFlatIn in{A};
    const int ngroups = hs_merge_small_core(in, A.spec, (int)n64, A.n_rows, A.n_order, A.cap, lds, A.out_rep, A.out_acc, err);
    if (threadIdx.x == 0) *A.out_ngroups = ngroups;
    if (err) atomicOr(A.flags, err);
}

// --------------------------------------------------------------------------------------------------
// The short tail, second half: slabs -> merged groups -> projection -> stored kinds -> result image.
struct FinishSink {
    const AggFinishArgs& A;
    const FinishView& V;
    int g;
    uint32_t& err;
    __device__ __forceinline__ FinishSink(const AggFinishArgs& a, const FinishView& v, uint32_t& e) : A(a), V(v), g(0), err(e) {}
    __device__ __forceinline__ void store(const hs_finish_out& d, uint64_t cell) {
        uint8_t* col = V.image + d.offset;
        switch (d.kind) {
            case HS_F32: {
                const double v = hs_u2d(cell);
                const float f = (float)v;
                if (isinf(f) && !isinf(v)) err |= HS_FLAG_FLT_OVERFLOW;
                ((float*)col)[g] = f;
                break;
            }
            case HS_I32: {
                const int64_t v = (int64_t)cell;
                if (v > 2147483647ll || v < -2147483648ll) err |= HS_FLAG_INT_OVERFLOW;
                ((int32_t*)col)[g] = (int32_t)v;
                break;
            }
            default: ((uint64_t*)col)[g] = cell; break;
        }
    }
    __device__ __forceinline__ uint64_t load(uint32_t s, int) const {
        const int j = A.fin.prog_src[s];
        if (j >= 0) return V.out_acc[(int64_t)j * A.cap + g];
        const hs_col kc{A.desc.key_kind, -1, V.key_scratch, nullptr, nullptr};
        return hs_load_cell(kc, g);
    }
    __device__ __forceinline__ bool live(int) const { return true; }
    __device__ __forceinline__ int64_t row(int) const { return g; }
    __device__ __forceinline__ void filter(int, bool) {}
    __device__ __forceinline__ void agg(uint32_t, int, uint64_t) {}
    __device__ __forceinline__ void key() {}
    __device__ __forceinline__ void out(uint32_t o, int, uint64_t cell) { store(A.fin.outs[A.fin.prog_out[o]], cell); }
};

// SMALL: the slabs, the merged cells and the result image all live in LDS (every global hop of this
// one-workgroup kernel is a serialised ~2 us round trip); otherwise they stay in global memory.
template <bool SMALL>
__global__ void __launch_bounds__(1024) k_agg_finish(const AggFinishArgs A_kernarg) {
    HS_KERNARG_LDS(AggFinishArgs, A);
    extern __shared__ __align__(16) uint64_t lds[];
    __shared__ uint32_t s_err, s_flags0;
    const int tid = threadIdx.x, nthr = blockDim.x;
    int64_t* const stamps = A.stamps;
#define HS_STAMP(i) do { if (stamps && tid == 0) stamps[i] = (int64_t)wall_clock64(); } while (0)
    HS_STAMP(0);
    const int n = A.world * (int)A.desc.slab_rows;
    uint8_t* const lds8 = (uint8_t*)lds;
    FinishView V;
    if constexpr (SMALL) {
        const uint64_t* src = (const uint64_t*)A.slabs;  // one parallel load brings every partial row on chip
        uint64_t* dst = (uint64_t*)(lds8 + A.lds_slabs);
        const int64_t words = (int64_t)A.world * A.desc.stride / 8;
        for (int64_t i = tid; i < words; i += nthr) dst[i] = src[i];
        V.slabs = lds8 + A.lds_slabs;
        V.out_rep = (int64_t*)(lds8 + A.lds_work);
        V.out_acc = (uint64_t*)(lds8 + A.lds_work) + A.cap;
        V.key_scratch = (uint8_t*)((uint64_t*)(lds8 + A.lds_work) + (int64_t)A.cap * (A.fin.n_fold + 1));
        V.image = lds8 + A.lds_image;
    } else {
        V.slabs = A.slabs;
        V.out_rep = A.out_rep;
        V.out_acc = A.out_acc;
        V.key_scratch = A.key_scratch;
        V.image = A.result;
    }
    if (tid == 0) {
        // status of everything that ran before this launch (final: the stream is ordered); fetched now so the
        // end of the kernel does not wait for another global round trip
        s_flags0 = __hip_atomic_load(A.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_err = 0;
    }
    __syncthreads();
    HS_STAMP(1);
    uint32_t err = 0;
    const MergeSlabIn in{A, V};
    const int ng = hs_merge_small_core(in, A.spec, n, n, A.n_order, A.cap, lds, V.out_rep, V.out_acc, err, stamps);
    __syncthreads();  // out_rep / out_acc are complete (one workgroup: visible after the barrier wherever they live)
    const int kb = A.key_bytes;
    for (int g = tid; g < ng; g += nthr) {
        int i;
        const uint8_t* s = in.slab_of((int)V.out_rep[g], i);
        const uint8_t* src = s + A.desc.key_off + (int64_t)i * kb;
        for (int b = 0; b < kb; ++b) V.key_scratch[(int64_t)g * kb + b] = src[b];
    }
    __syncthreads();
    HS_STAMP(8);
    FinishSink sink(A, V, err);
    for (int t = tid; t < A.fin.n_out * ng; t += nthr) {  // key and merged aggregates that pass through
        const int o = t / ng, g = t - o * ng;
        const hs_finish_out& d = A.fin.outs[o];
        if (d.src == 0) {
            for (int b = 0; b < kb; ++b) V.image[d.offset + (int64_t)g * kb + b] = V.key_scratch[(int64_t)g * kb + b];
        } else if (d.src == 1) {
            sink.g = g;
            sink.store(d, V.out_acc[(int64_t)d.index * A.cap + g]);
        }
    }
    HS_STAMP(9);
    if (A.n_simple > 0) {
        for (int t = tid; t < A.n_simple * ng; t += nthr) {
            const int i = t / ng;
            sink.g = t - i * ng;
            uint64_t x = sink.load((uint32_t)A.simple[i].a, 0), y = sink.load((uint32_t)A.simple[i].b, 0);
            if (A.simple[i].conv_a) x = hs_d2u((double)(int64_t)x);
            if (A.simple[i].conv_b) y = hs_d2u((double)(int64_t)y);
            uint64_t v;
            switch (A.simple[i].op) {
                case HS_OP_ADD_F: v = hs_bin<HS_OP_ADD_F>(x, y, true, err); break;
                case HS_OP_SUB_F: v = hs_bin<HS_OP_SUB_F>(x, y, true, err); break;
                case HS_OP_MUL_F: v = hs_bin<HS_OP_MUL_F>(x, y, true, err); break;
                case HS_OP_DIV_F: v = hs_bin<HS_OP_DIV_F>(x, y, true, err); break;
                case HS_OP_ADD_I: v = hs_bin<HS_OP_ADD_I>(x, y, true, err); break;
                case HS_OP_SUB_I: v = hs_bin<HS_OP_SUB_I>(x, y, true, err); break;
                default: v = hs_bin<HS_OP_MUL_I>(x, y, true, err); break;
            }
            sink.store(A.fin.outs[A.simple[i].out], v);
        }
    } else if (A.prog.n_ins && tid < HS_WAVE) {  // the projection: one wave, interpreter stack in LDS
        uint64_t* stack = (uint64_t*)(lds8 + A.lds_stack) + tid;
        for (int g = tid; g < ng; g += HS_WAVE) {
            sink.g = g;
            hs_run_compact(A.prog, A.nocols, sink, err, stack, HS_WAVE);
        }
    }
    HS_STAMP(10);
    if (err) atomicOr(&s_err, err);
    __syncthreads();
    if (tid == 0) {
        uint32_t f = s_flags0 | s_err;
        for (int k = 0; k < A.world; ++k) f |= *(const uint32_t*)(V.slabs + (int64_t)k * A.desc.stride);
        ((uint32_t*)V.image)[0] = f;
        ((int64_t*)V.image)[1] = ng;
        *A.flags = 0;  // status handed over: the words are clean for the next run
        if (A.own_slab_flags) *A.own_slab_flags = 0;
    }
    if constexpr (SMALL) {  // staged image -> result, everything but the "done" word
        __syncthreads();
        const uint32_t* src = (const uint32_t*)V.image;
        uint32_t* dst = (uint32_t*)A.result;
        for (int64_t i = tid; i < A.image_bytes / 4; i += nthr) {
            if (i != 1) dst[i] = src[i];
        }
    }
    HS_STAMP(11);
    __threadfence_system();  // the result may sit in mapped host memory: every byte visible before "done"
    __syncthreads();
    HS_STAMP(12);
    if (tid == 0) __hip_atomic_store((uint32_t*)A.result + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#undef HS_STAMP
}

// ==================================================================================================
// host launchers (C ABI)
// ==================================================================================================
extern thread_local char g_hs_err[256];
void hs_set_error(const char* fmt, ...);
// hs_jit.cpp: HS_OK = a kernel compiled for this program was launched; anything else = not launched
int hs_jit_launch_agg_main(const AggMainArgs* args, bool hashed, unsigned grid, unsigned block, size_t lds_bytes,
                           hipStream_t stream);

static int fill_cols(HsCols& dst, const hs_col* cols, int32_t n) {
    if (n < 0 || n > HS_MAX_COLS) {
        hs_set_error("n_cols=%d exceeds HS_MAX_COLS=%d", n, HS_MAX_COLS);
        return HS_E_LIMIT;
    }
    dst.n = n;
    dst.pad = 0;
    for (int i = 0; i < n; ++i) dst.c[i] = cols[i];
    for (int i = n; i < HS_MAX_COLS; ++i) dst.c[i] = hs_col{HS_U8, -1, nullptr, nullptr, nullptr};
    return HS_OK;
}

static int program_depth(const hs_program* p) {
    int d = 0;
    for (uint32_t i = 0; i < p->n_ins; ++i) {
        int sp = (int)((p->ins[i] >> 8) & 0xff);
        int op = (int)(p->ins[i] & 0xff);
        int after = sp;
        if (op == HS_OP_LD || op == HS_OP_LIT || op == HS_OP_STRCMP_LIT || op == HS_OP_STRCMP_COL || op == HS_OP_LIKE ||
            op == HS_OP_DICTBIT)
            after = sp + 1;
        if (after > d) d = after;
        if (sp > d) d = sp;
    }
    return d;
}

static constexpr size_t HS_LDS_SOFT = 64 * 1024;   // keeps >= 2 workgroups per CU resident
static constexpr size_t HS_LDS_HARD = 144 * 1024;  // one workgroup per CU (gfx950: 160 KiB per CU)

// hs_agg_geom.pad of the private-table tier: set when every unit owns at least one chunk (fused unit combine allowed)
static constexpr int32_t HS_GEOM_FUSABLE = 0x46555345;  // "FUSE"
static size_t hs_agg_partials_bytes(int64_t n_chunks, int32_t group_cap, int32_t n_acc) {
    const size_t slots = (size_t)n_chunks * (size_t)group_cap;
    return (slots * 16 + slots * (size_t)n_acc * 8 + 256 + 15) & ~(size_t)15;
}

static size_t agg_main_lds(int32_t group_cap, int32_t n_acc, int wg) {
    return (size_t)group_cap * 16 + (size_t)group_cap * (size_t)n_acc * (size_t)wg * 8;
}

extern "C" int hs_agg_partial_geom(const int64_t* host_unit_rows, int64_t n_units, int32_t n_acc, int32_t group_cap,
                                   hs_agg_geom* out) {
    if (!host_unit_rows || !out || n_units < 0 || n_acc < 0 || n_acc > HS_MAX_ACC || group_cap < 1 ||
        (group_cap & (group_cap - 1))) {
        hs_set_error("hs_agg_partial_geom: bad arguments");
        return HS_E_ARG;
    }
    // every lane owns a private [group_cap][n_acc] table in LDS: pick the widest workgroup that fits
    int wg = 0;
    size_t lds = 0;
    const int widths[3] = {256, 128, 64};
    for (int pass = 0; pass < 2 && !wg; ++pass) {
        for (int k = 0; k < 3; ++k) {
            lds = agg_main_lds(group_cap, n_acc, widths[k]);
            if (lds <= (pass == 0 ? HS_LDS_SOFT : HS_LDS_HARD)) {
                wg = widths[k];
                break;
            }
        }
    }
    if (!wg) {
        hs_set_error("hs_agg_partial_geom: group_cap=%d x n_acc=%d private tables need %zu B LDS even at 64 lanes (> %zu)",
                     group_cap, n_acc, agg_main_lds(group_cap, n_acc, 64), HS_LDS_HARD);
        return HS_E_LIMIT;
    }
    int64_t total = 0;
    for (int64_t u = 0; u < n_units; ++u) {
        const int64_t r = host_unit_rows[u + 1] - host_unit_rows[u];
        if (r < 0) {
            hs_set_error("hs_agg_partial_geom: unit_rows not ascending at %lld", (long long)u);
            return HS_E_ARG;
        }
        total += r;
    }
    // Rows per workgroup = steps_per_chunk * wg * V.  Model from a chunk-size sweep on MI355X:
    // a workgroup costs ~11 us fixed + ~3.4 us per step, and `wg_slots` workgroups run at once (LDS-limited).
    // Pick the number of full-chip rounds R that keeps chunks <= 128 steps, then the chunk length that
    // fills those rounds exactly: few, long chunks without a ragged last round.
    const int64_t step = (int64_t)wg * HS_V;
    static int n_cus = 0;
    if (!n_cus) {
        hipDeviceProp_t prop;
        int dev = 0;
        n_cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                 prop.multiProcessorCount > 0)
                    ? prop.multiProcessorCount
                    : 256;
        (void)hipGetLastError();
    }
    int64_t per_cu = (int64_t)(160 * 1024) / (int64_t)(lds > 1 ? lds : 1);
    const int64_t by_waves = 32 / (wg / 64);
    if (per_cu > by_waves) per_cu = by_waves;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    const int64_t wg_slots = (int64_t)n_cus * per_cu;
    const int64_t total_steps = (total + step - 1) / step;
    int64_t steps_per_chunk;
    if (const char* e = getenv("HIPSPARK_CHUNK_STEPS")) {  // tuning knob
        steps_per_chunk = atol(e);
    } else {
        const int64_t max_steps = 128;
        int64_t rounds = (total_steps + wg_slots * max_steps - 1) / (wg_slots * max_steps);
        if (rounds < 1) rounds = 1;
        steps_per_chunk = (total_steps + wg_slots * rounds - 1) / (wg_slots * rounds);
        if (steps_per_chunk < 1) steps_per_chunk = 1;
        // every unit rounds its chunk count up: lengthen the chunks until the grid really fits the rounds
        // (a handful of workgroups spilling into an extra round would cost a whole round)
        for (int guard = 0; guard < 64; ++guard) {
            int64_t n = 0;
            for (int64_t u = 0; u < n_units; ++u) {
                const int64_t span = host_unit_rows[u + 1] - (host_unit_rows[u] & ~(int64_t)(HS_V - 1));
                n += span > 0 ? (span + step * steps_per_chunk - 1) / (step * steps_per_chunk) : 0;
            }
            if (n <= wg_slots * rounds) break;
            ++steps_per_chunk;
        }
        // small tables: a workgroup costs ~11 us before its first and after its last step, and the unit combine in the last
        // one's epilogue walks the unit's chunks in batches of 64 - filling every workgroup slot with one- or two-step chunks
        // is slower than fewer, longer ones as long as every CU still gets one.  Measured at sf=1 (6 M rows, 3 units;
        // profiles/r04_chunk_steps_sweep.txt): 8 steps (733 chunks) 94 us, 16 steps (367) 73 us, 32 steps (184) 82 us.
        int64_t floor = total_steps / n_cus;
        floor = floor < 1 ? 1 : (floor > 16 ? 16 : floor);
        if (steps_per_chunk < floor) steps_per_chunk = floor;
    }
    if (steps_per_chunk < 1) steps_per_chunk = 1;
    if (steps_per_chunk > 4096) steps_per_chunk = 4096;
    const int64_t chunk = step * steps_per_chunk;
    int64_t n_chunks = 0;
    for (int64_t u = 0; u < n_units; ++u) {
        const int64_t anchor = host_unit_rows[u] & ~(int64_t)(HS_V - 1);
        const int64_t span = host_unit_rows[u + 1] - anchor;
        n_chunks += span > 0 ? (span + chunk - 1) / chunk : 0;
    }
    bool every_unit_has_rows = n_units > 0;
    for (int64_t u = 0; u < n_units; ++u) every_unit_has_rows = every_unit_has_rows && host_unit_rows[u + 1] > host_unit_rows[u];
    out->group_cap = group_cap;
    out->chunk_rows = (int32_t)chunk;
    out->wg_threads = wg;
    // a unit without rows owns no chunk, so no workgroup would ever combine it: such launches keep the separate
    // combine kernel (which writes the empty unit's outputs)
    out->pad = every_unit_has_rows ? HS_GEOM_FUSABLE : 0;
    out->n_chunks = n_chunks;
    out->lds_bytes = lds;
    // chunk partials + 256 spare bytes + one arrival counter per unit (fused combine)
    out->ws_bytes = hs_agg_partials_bytes(n_chunks, group_cap, n_acc) + (size_t)n_units * 4 + 64;
    return HS_OK;
}

extern "C" int hs_agg_partial_chunks(const int64_t* host_unit_rows, int64_t n_units, const hs_agg_geom* geom,
                                     hs_chunk* host_chunks, int64_t* host_unit_chunk0) {
    if (!host_unit_rows || !geom || !host_chunks || !host_unit_chunk0 || n_units < 0 || geom->chunk_rows < HS_V) {
        hs_set_error("hs_agg_partial_chunks: bad arguments");
        return HS_E_ARG;
    }
    int64_t k = 0;
    for (int64_t u = 0; u < n_units; ++u) {
        host_unit_chunk0[u] = k;
        const int64_t us = host_unit_rows[u], ue = host_unit_rows[u + 1];
        for (int64_t c0 = us & ~(int64_t)(HS_V - 1); c0 < ue; c0 += geom->chunk_rows) {
            if (k >= geom->n_chunks) {
                hs_set_error("hs_agg_partial_chunks: geometry does not match the unit boundaries");
                return HS_E_ARG;
            }
            const int64_t c1 = c0 + geom->chunk_rows < ue ? c0 + geom->chunk_rows : ue;
            host_chunks[k++] = hs_chunk{c0, c1, us, u};
        }
    }
    host_unit_chunk0[n_units] = k;
    if (k != geom->n_chunks) {
        hs_set_error("hs_agg_partial_chunks: geometry does not match the unit boundaries");
        return HS_E_ARG;
    }
    return HS_OK;
}

template <typename K>
static void allow_big_lds(K kernel) {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HS_LDS_HARD);
}

// debug: per-workgroup phase stamps of the private-table scan (HIPSPARK_SCAN_STAMPS=1, tools/scan_stamps.py).  The
// buffer belongs to the library and is only ever grown, so a captured launch keeps a valid pointer.
static int64_t* g_scan_stamps = nullptr;
static int64_t g_scan_stamps_cap = 0, g_scan_stamps_chunks = 0;
static int64_t* hs_scan_stamps_for(int64_t n_chunks) {
    static const bool want = getenv("HIPSPARK_SCAN_STAMPS") && getenv("HIPSPARK_SCAN_STAMPS")[0] == '1';
    if (!want) return nullptr;
    if (n_chunks > g_scan_stamps_cap) {
        int64_t* fresh = nullptr;
        if (hipMalloc((void**)&fresh, (size_t)n_chunks * 128) != hipSuccess) return nullptr;
        g_scan_stamps = fresh;  // (the old block stays allocated: an earlier captured launch may still name it)
        g_scan_stamps_cap = n_chunks;
    }
    g_scan_stamps_chunks = n_chunks;
    return g_scan_stamps;
}
extern "C" int64_t hs_agg_debug_scan_stamps(int64_t* host_out, int64_t max_chunks) {
    if (!g_scan_stamps || !host_out) return 0;
    const int64_t n = g_scan_stamps_chunks < max_chunks ? g_scan_stamps_chunks : max_chunks;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host_out, g_scan_stamps, (size_t)n * 128, hipMemcpyDeviceToHost) != hipSuccess) {
        hs_set_error("hs_agg_debug_scan_stamps: copy failed");
        return -1;
    }
    return n;
}

static int agg_partial_impl(void* stream, const hs_col* cols, int32_t n_cols, int32_t key_col, const hs_program* prog,
                            const hs_agg_spec* spec, const hs_chunk* chunks, const int64_t* unit_chunk0,
                            int64_t n_units, const hs_agg_geom* geom, int64_t* out_rep, uint64_t* out_acc,
                            int32_t* out_ngroups, const int64_t* unit_ids, uint8_t* slab, const hs_slab_desc* desc,
                            void* ws, uint32_t* flags, void* ev_begin, void* ev_end) {
    if (!cols || !prog || !spec || !chunks || !unit_chunk0 || !geom || !ws || !flags || key_col < 0 ||
        key_col >= n_cols || (!slab && (!out_rep || !out_acc || !out_ngroups)) || (slab && !desc)) {
        hs_set_error("hs_agg_partial: null or out-of-range argument");
        return HS_E_ARG;
    }
    int key_bytes = 0;
    if (slab) {
        const hs_col& kc = cols[key_col];
        if (kc.kind == HS_STR) key_bytes = (kc.fixed_len == 1 || kc.fixed_len == 2 || kc.fixed_len == 4) ? kc.fixed_len : 0;
        else key_bytes = kc.kind == HS_U8 ? 1 : (kc.kind == HS_I32 || kc.kind == HS_F32) ? 4 : 8;
        if (!key_bytes || desc->key_kind != kc.kind || (kc.kind == HS_STR && desc->key_len != kc.fixed_len) ||
            desc->n_acc != spec->n_acc || n_units * (int64_t)geom->group_cap > desc->slab_rows) {
            hs_set_error("hs_agg_partial_slab: slab description does not match the key column / aggregates / units");
            return HS_E_ARG;
        }
        for (int a = 0; a < spec->n_acc; ++a) {
            if (desc->acc_kind[a] != (spec->is_int[a] ? HS_I32 : HS_F32)) {
                hs_set_error("hs_agg_partial_slab: accumulator %d kind mismatch", a);
                return HS_E_ARG;
            }
        }
    }
    if (prog->n_ins > HS_MAX_INS || prog->n_lit > HS_MAX_LIT) {
        hs_set_error("hs_agg_partial: program too long");
        return HS_E_LIMIT;
    }
    if (n_units == 0 || geom->n_chunks == 0) return HS_OK;
    AggMainArgs A;
    int rc = fill_cols(A.cols, cols, n_cols);
    if (rc) return rc;
    // numeric columns and a preloaded key must sit in the first HS_FUSED_COLS slots
    for (int i = HS_FUSED_COLS; i < n_cols; ++i) {
        if (cols[i].kind != HS_STR || i == key_col) {
            hs_set_error("hs_agg_partial: more than %d numeric column slots", HS_FUSED_COLS);
            return HS_E_LIMIT;
        }
    }
    A.prog = *prog;
    A.spec = *spec;
    A.key_col = key_col;
    A.group_cap = geom->group_cap;
    A.chunk_rows = geom->chunk_rows;
    A.pad = 0;
    A.chunks = chunks;
    A.unit_chunk0 = unit_chunk0;
    A.n_units = n_units;
    const size_t slots = (size_t)geom->n_chunks * geom->group_cap;
    A.part_keys = (uint64_t*)ws;
    A.part_rep = (int64_t*)((char*)ws + slots * 8);
    A.part_acc = (uint64_t*)((char*)ws + slots * 16);
    A.flags = flags;
    A.replicas = 0;
    A.pad2 = 0;
    const bool hashed = !hs_col_packs(cols[key_col]);
    const int depth = program_depth(prog);
    if (depth > HS_MAX_STACK) {
        hs_set_error("hs_agg_partial: expression stack depth %d > %d", depth, HS_MAX_STACK);
        return HS_E_LIMIT;
    }
    hipStream_t s = (hipStream_t)stream;
    if (geom->wg_threads != 64 && geom->wg_threads != 128 && geom->wg_threads != 256) {
        hs_set_error("hs_agg_partial: geometry not made by hs_agg_partial_geom");
        return HS_E_ARG;
    }
    dim3 grid((unsigned)geom->n_chunks), block((unsigned)geom->wg_threads);
    static unsigned long long attrs_set = 0;
    if (hs_first_on_device(attrs_set)) {
        allow_big_lds(k_agg_main<true, 8>);
        allow_big_lds(k_agg_main<false, 4>);
        allow_big_lds(k_agg_main<false, 8>);
    }
    if (geom->n_chunks > 0x7fffffffll) {
        hs_set_error("hs_agg_partial: too many chunks");
        return HS_E_LIMIT;
    }
    // ---- the unit combine: arguments first, then decide whether it rides in the scan kernel's epilogue ----
    AggUnitArgs U;
    U.key = cols[key_col];
    U.spec = *spec;
    U.group_cap = geom->group_cap;
    U.hashed = hashed ? 1 : 0;
    U.unit_chunk0 = unit_chunk0;
    U.part_keys = A.part_keys;
    U.part_rep = A.part_rep;
    U.part_acc = A.part_acc;
    U.out_rep = out_rep;
    U.out_acc = out_acc;
    U.out_ngroups = out_ngroups;
    U.flags = flags;
    U.slab = slab;
    U.unit_ids = unit_ids;
    U.order_off = U.key_off = 0;
    U.key_bytes = key_bytes;
    U.pad2 = 0;
    for (int a = 0; a < HS_MAX_ACC; ++a) U.acc_off[a] = 0;
    if (slab) {
        U.order_off = desc->order_off;
        U.key_off = desc->key_off;
        for (int a = 0; a < spec->n_acc; ++a) U.acc_off[a] = desc->acc_off[a];
    }
    const size_t ubase = (size_t)geom->group_cap * 16 + (size_t)geom->group_cap * spec->n_acc * 16;  // (+ the staged batch's spare column)
    const size_t uper = (size_t)geom->group_cap * spec->n_acc * 8 + (size_t)geom->group_cap * 4;  // per staged chunk
    int ubatch = HS_UNIT_BATCH;
    while (ubatch > 1 && ubase + (size_t)ubatch * uper > HS_LDS_SOFT) ubatch /= 2;
    const size_t ulds = ubase + (size_t)ubatch * uper;
    if (ulds > HS_LDS_HARD) {
        hs_set_error("hs_agg_partial: unit combine needs %zu B LDS", ulds);
        return HS_E_LIMIT;
    }
    U.batch = ubatch;
    U.pad = 0;
    // Fused form: the workgroup that finishes a unit's last chunk combines the unit inside the scan
    // kernel, staging through the kernel's own LDS block.  Needs: the arrival counters (tail of `ws`, zero before the
    // first launch, left zero by every launch), every unit owning at least one chunk (the caller says so by passing
    // fused_ok), a staging batch that fits that block.
    A.unit_arrivals = nullptr;
    A.unit = U;
    A.unit_col = -1;
    A.pad3 = 0;
    A.chunk_acc = nullptr;
    A.stamps = hs_scan_stamps_for(geom->n_chunks);
    bool fused = false;
    if (geom->pad == HS_GEOM_FUSABLE && spec->n_acc > 0) {
        int fbatch = ubatch;
        while (fbatch > 1 && ubase + (size_t)fbatch * uper > geom->lds_bytes) fbatch /= 2;
        if (ubase + (size_t)fbatch * uper <= geom->lds_bytes) {
            A.unit.batch = fbatch;
            A.unit_arrivals = (uint32_t*)((char*)ws + hs_agg_partials_bytes(geom->n_chunks, geom->group_cap, spec->n_acc));
            fused = true;
        }
    }
    if (ev_begin) hs_event_record((hipEvent_t)ev_begin, s);
    const int jit_rc = hs_jit_launch_agg_main(&A, hashed, grid.x, block.x, geom->lds_bytes, s);
    if (jit_rc == HS_OK) {
        // launched the program compiled for exactly this bytecode
    } else if (hashed) {
        hipLaunchKernelGGL((k_agg_main<true, 8>), grid, block, geom->lds_bytes, s, A);
    } else if (depth <= 4) {
        hipLaunchKernelGGL((k_agg_main<false, 4>), grid, block, geom->lds_bytes, s, A);
    } else {
        hipLaunchKernelGGL((k_agg_main<false, 8>), grid, block, geom->lds_bytes, s, A);
    }
    if (ev_end) hs_event_record((hipEvent_t)ev_end, s);
    if (!fused) {
        static unsigned long long unit_attr = 0;
        if (hs_first_on_device(unit_attr)) allow_big_lds(k_agg_unit);
        hipLaunchKernelGGL(k_agg_unit, dim3((unsigned)n_units), dim3(256), ulds, s, U);
    }
    if (hipGetLastError() != hipSuccess) {
        hs_set_error("hs_agg_partial: kernel launch failed");
        return HS_E_LAUNCH;
    }
    return HS_OK;
}

extern "C" int hs_agg_partial(void* stream, const hs_col* cols, int32_t n_cols, int32_t key_col,
                              const hs_program* prog, const hs_agg_spec* spec, const hs_chunk* chunks,
                              const int64_t* unit_chunk0, int64_t n_units, const hs_agg_geom* geom, int64_t* out_rep,
                              uint64_t* out_acc, int32_t* out_ngroups, void* ws, uint32_t* flags, void* ev_begin,
                              void* ev_end) {
    return agg_partial_impl(stream, cols, n_cols, key_col, prog, spec, chunks, unit_chunk0, n_units, geom, out_rep,
                            out_acc, out_ngroups, nullptr, nullptr, nullptr, ws, flags, ev_begin, ev_end);
}

extern "C" int hs_agg_partial_slab(void* stream, const hs_col* cols, int32_t n_cols, int32_t key_col,
                                   const hs_program* prog, const hs_agg_spec* spec, const hs_chunk* chunks,
                                   const int64_t* unit_chunk0, int64_t n_units, const hs_agg_geom* geom,
                                   const int64_t* unit_ids, uint8_t* slab, const hs_slab_desc* desc, void* ws,
                                   uint32_t* flags, void* ev_begin, void* ev_end) {
    if (!slab || !desc) {
        hs_set_error("hs_agg_partial_slab: null slab");
        return HS_E_ARG;
    }
    return agg_partial_impl(stream, cols, n_cols, key_col, prog, spec, chunks, unit_chunk0, n_units, geom, nullptr,
                            nullptr, nullptr, unit_ids, slab, desc, ws, flags, ev_begin, ev_end);
}

extern "C" int hs_agg_pack(void* stream, const int64_t* rep, const uint64_t* acc, const int32_t* ngroups,
                           int64_t n_units, int32_t group_cap, const hs_agg_spec* spec, int64_t* pack_start,
                           int64_t* out_rep, void* const* out_cols, const int32_t* acc_kinds, const int64_t* unit_ids,
                           int64_t* out_unit) {
    if (!rep || !acc || !ngroups || !spec || !pack_start || !out_rep || (spec->n_acc > 0 && (!out_cols || !acc_kinds))) {
        hs_set_error("hs_agg_pack: null argument");
        return HS_E_ARG;
    }
    AggPackArgs A;
    A.rep = rep;
    A.acc = acc;
    A.ngroups = ngroups;
    A.n_units = n_units;
    A.group_cap = group_cap;
    A.n_acc = spec->n_acc;
    A.pack_start = pack_start;
    A.out_rep = out_rep;
    A.unit_ids = unit_ids;
    A.out_unit = out_unit;
    for (int a = 0; a < HS_MAX_ACC; ++a) {
        A.out_cols[a] = a < spec->n_acc ? out_cols[a] : nullptr;
        A.acc_kinds[a] = a < spec->n_acc ? acc_kinds[a] : HS_F64;
    }
    if (group_cap > 64 && n_units > 0) {  // wide units: one workgroup per unit instead of one lane per unit
        hipLaunchKernelGGL(k_agg_pack_scan, dim3(1), dim3(256), 0, (hipStream_t)stream, ngroups, n_units, pack_start);
        hipLaunchKernelGGL(k_agg_pack_wide, dim3((unsigned)n_units), dim3(256), 0, (hipStream_t)stream, A);
    } else {
        hipLaunchKernelGGL(k_agg_pack, dim3(1), dim3(256), 0, (hipStream_t)stream, A);
    }
    if (hipGetLastError() != hipSuccess) {
        hs_set_error("hs_agg_pack: kernel launch failed");
        return HS_E_LAUNCH;
    }
    return HS_OK;
}

static constexpr size_t HS_MERGE_LDS_MAX = 150 * 1024;  // dynamic part; the kernels add < 4 KB static
// threads of the single-workgroup merge: barriers cost per wave, so small inputs get a small workgroup
static unsigned merge_block(int64_t n_rows, int cap, int n_order) {
    const int64_t m = n_rows > cap ? (n_rows > n_order ? n_rows : n_order) : (cap > n_order ? cap : n_order);
    return m <= 256 ? 256u : m <= 512 ? 512u : 1024u;
}

extern "C" int hs_agg_merge(void* stream, const hs_col* key, const hs_col* acc_cols, const hs_agg_spec* spec,
                            const int64_t* order, int64_t n_order, int64_t n_rows, const int64_t* n_rows_dev,
                            int32_t cap, int64_t* out_rep, uint64_t* out_acc, int64_t* out_ngroups, uint32_t* flags) {
    if (!key || !spec || !out_rep || !out_acc || !out_ngroups || !flags || cap < 1 || (cap & (cap - 1)) ||
        (spec->n_acc > 0 && !acc_cols)) {
        hs_set_error("hs_agg_merge: bad arguments");
        return HS_E_ARG;
    }
    if (order && (n_order < 1 || n_order > 65536)) {
        hs_set_error("hs_agg_merge: n_order=%lld out of range", (long long)n_order);
        return HS_E_ARG;
    }
    const size_t fixed = (size_t)cap * 32 + (order ? (size_t)n_order * 8 : 0) + 16, per_row = (size_t)spec->n_acc * 8 + 12;
    size_t lds = fixed + (size_t)n_rows * per_row;
    int32_t limited = 0;
    if (lds > HS_MERGE_LDS_MAX) {
        // n_rows is an upper bound when the real count lives on the device (dense partial rows of the shared tier:
        // units x table capacity): run with as many rows as LDS holds; the kernel raises HS_FLAG_MERGE_ROWS when the
        // device count is larger, and the caller takes the HBM-tier merge from then on
        const int64_t fit = fixed < HS_MERGE_LDS_MAX ? (int64_t)((HS_MERGE_LDS_MAX - fixed) / per_row) : 0;
        if (!n_rows_dev || fit < 64) {
            hs_set_error("hs_agg_merge: %lld partial rows x %d accumulators need %zu B LDS (> %zu): use the HBM-tier merge",
                         (long long)n_rows, spec->n_acc, lds, HS_MERGE_LDS_MAX);
            return HS_E_LIMIT;
        }
        n_rows = fit;
        limited = 1;
        lds = fixed + (size_t)n_rows * per_row;
    }
    AggMergeArgs A;
    A.key = *key;
    for (int a = 0; a < HS_MAX_ACC; ++a)
        A.acc_cols[a] = a < spec->n_acc ? acc_cols[a] : hs_col{HS_U8, -1, nullptr, nullptr, nullptr};
    A.spec = *spec;
    A.n_rows = n_rows;
    A.n_rows_dev = n_rows_dev;
    A.order = order;
    A.n_order = order ? (int32_t)n_order : 0;
    A.limited = limited;
    A.cap = cap;
    A.hashed = hs_col_packs(*key) ? 0 : 1;
    A.out_rep = out_rep;
    A.out_acc = out_acc;
    A.out_ngroups = out_ngroups;
    A.flags = flags;
    static unsigned long long attr_set = 0;
    if (hs_first_on_device(attr_set)) {
        hipFuncSetAttribute((const void*)k_agg_merge_small, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)HS_MERGE_LDS_MAX);
    }
    hipLaunchKernelGGL(k_agg_merge_small, dim3(1), dim3(merge_block(n_rows, cap, A.n_order)), lds, (hipStream_t)stream, A);
    if (hipGetLastError() != hipSuccess) {
        hs_set_error("hs_agg_merge: kernel launch failed");
        return HS_E_LAUNCH;
    }
    return HS_OK;
}

extern "C" size_t hs_agg_finish_scratch_bytes(int32_t cap, int32_t n_fold) {
    return (size_t)cap * 8 * (size_t)(n_fold + 2) + 64 + 128;  // out_rep + out_acc + key elements (<= 8 bytes each) + stamps
}

extern "C" int hs_agg_finish(void* stream, const uint8_t* gathered, int32_t world, const hs_slab_desc* desc,
                             const hs_finish_spec* fin, const hs_program* prog, int64_t n_order, int32_t cap,
                             uint8_t* result, void* scratch, uint32_t* flags, uint32_t* own_slab_flags) {
    if (!gathered || !desc || !fin || !result || !scratch || !flags || world < 1 || cap < 1 || (cap & (cap - 1)) ||
        fin->n_fold < 0 || fin->n_fold > HS_MAX_ACC || fin->n_out < 0 || fin->n_out > HS_FINISH_MAX_OUT ||
        desc->slab_rows < 1 || desc->n_acc < 0 || desc->n_acc > HS_MAX_ACC) {
        hs_set_error("hs_agg_finish: bad arguments");
        return HS_E_ARG;
    }
    if (n_order < 1 || n_order > 65536) {
        hs_set_error("hs_agg_finish: n_order=%lld out of range", (long long)n_order);
        return HS_E_ARG;
    }
    int key_bytes;
    if (desc->key_kind == HS_STR) {
        if (desc->key_len != 1 && desc->key_len != 2 && desc->key_len != 4) {
            hs_set_error("hs_agg_finish: string keys must have a fixed length of 1, 2 or 4 bytes");
            return HS_E_ARG;
        }
        key_bytes = desc->key_len;
    } else {
        key_bytes = desc->key_kind == HS_U8 ? 1 : (desc->key_kind == HS_I32 || desc->key_kind == HS_F32) ? 4 : 8;
    }
    const int64_t n_rows = (int64_t)world * desc->slab_rows;
    const size_t lds = (size_t)cap * 16 + (size_t)n_rows * fin->n_fold * 8 + (size_t)n_rows * 12 + (size_t)cap * 16 +
                       (size_t)n_order * 8 + 16;
    if (n_rows > 0x3fffffff || lds > HS_MERGE_LDS_MAX) {
        hs_set_error("hs_agg_finish: %lld partial rows x %d aggregates need %zu B LDS (> %zu)", (long long)n_rows,
                     fin->n_fold, lds, HS_MERGE_LDS_MAX);
        return HS_E_LIMIT;
    }
    AggFinishArgs A;
    memset(&A, 0, sizeof(A));
    A.slabs = gathered;
    A.desc = *desc;
    A.fin = *fin;
    A.spec.n_acc = fin->n_fold;
    for (int j = 0; j < fin->n_fold; ++j) {
        const int src = fin->fold_src[j];
        if (src < 0 || src >= desc->n_acc || (fin->fold_op[j] != HS_AGG_SUM && fin->fold_op[j] != HS_AGG_MIN &&
                                             fin->fold_op[j] != HS_AGG_MAX)) {
            hs_set_error("hs_agg_finish: fold %d is malformed", j);
            return HS_E_ARG;
        }
        A.spec.op[j] = (uint8_t)fin->fold_op[j];
        A.spec.is_int[j] = desc->acc_kind[src] == HS_I32 ? 1 : 0;
    }
    int n_prog_out = 0;
    for (int o = 0; o < fin->n_out; ++o) {
        const hs_finish_out& d = fin->outs[o];
        const bool ok = d.src == 0 || (d.src == 1 && d.index >= 0 && d.index < fin->n_fold) ||
                        (d.src == 2 && d.index >= 0 && d.index < HS_MAX_OUTS);
        if (!ok || d.offset < 16 || (d.src != 0 && d.kind != HS_F32 && d.kind != HS_I32 && d.kind != HS_I64)) {
            hs_set_error("hs_agg_finish: output %d is malformed", o);
            return HS_E_ARG;
        }
        if (d.src == 2) ++n_prog_out;
    }
    if (prog && prog->n_ins) {
        if (prog->n_ins > HS_MAX_INS || prog->n_lit > HS_MAX_LIT || program_depth(prog) > HS_MAX_STACK) {
            hs_set_error("hs_agg_finish: program too long or too deep");
            return HS_E_LIMIT;
        }
        A.prog = *prog;
        // decode "LD a; LD b; [I2F]; [I2F]; <binary op>; OUT o" sequences (see AggFinishArgs::simple)
        {
            int n_simple = 0;
            bool ok = true;
            uint32_t pc = 0;
            while (ok && pc < prog->n_ins) {
                int slot[2] = {-1, -1}, conv[2] = {0, 0}, depth = 0;
                for (; pc < prog->n_ins && ok; ++pc) {
                    const uint64_t w = prog->ins[pc];
                    const int op = (int)(w & 0xff), a = (int)((w >> 16) & 0xffff);
                    if (op == HS_OP_LD && depth < 2) slot[depth++] = a;
                    else if (op == HS_OP_I2F && depth == 2) conv[a ? 0 : 1] = 1;
                    else if (op == HS_OP_I2F && depth == 1 && a == 0) conv[0] = 1;
                    else break;
                }
                if (!ok || depth != 2 || pc + 1 >= prog->n_ins) { ok = false; break; }
                const int op = (int)(prog->ins[pc] & 0xff);
                const uint64_t out_w = prog->ins[pc + 1];
                const bool arith = op == HS_OP_ADD_F || op == HS_OP_SUB_F || op == HS_OP_MUL_F || op == HS_OP_DIV_F ||
                                   op == HS_OP_ADD_I || op == HS_OP_SUB_I || op == HS_OP_MUL_I;
                const int o = (int)((out_w >> 16) & 0xffff);
                if (!arith || (int)(out_w & 0xff) != HS_OP_OUT || o < 0 || o >= HS_MAX_OUTS || n_simple >= HS_MAX_OUTS ||
                    slot[0] >= HS_MAX_COLS || slot[1] >= HS_MAX_COLS) { ok = false; break; }
                A.simple[n_simple].out = (int16_t)fin->prog_out[o];
                A.simple[n_simple].op = (int16_t)op;
                A.simple[n_simple].a = (int16_t)slot[0];
                A.simple[n_simple].b = (int16_t)slot[1];
                A.simple[n_simple].conv_a = (int8_t)conv[0];
                A.simple[n_simple].conv_b = (int8_t)conv[1];
                ++n_simple;
                pc += 2;
            }
            A.n_simple = ok ? n_simple : 0;
        }
    } else if (n_prog_out) {
        hs_set_error("hs_agg_finish: program outputs without a program");
        return HS_E_ARG;
    }
    A.nocols.n = 0;
    for (int i = 0; i < HS_MAX_COLS; ++i) A.nocols.c[i] = hs_col{HS_U8, -1, nullptr, nullptr, nullptr};
    A.world = world;
    A.cap = cap;
    A.n_order = (int32_t)n_order;
    A.key_bytes = key_bytes;
    A.result = result;
    A.out_rep = (int64_t*)scratch;
    A.out_acc = (uint64_t*)scratch + cap;
    A.key_scratch = (uint8_t*)((uint64_t*)scratch + (size_t)cap * (fin->n_fold + 1));
    A.flags = flags;
    A.own_slab_flags = own_slab_flags;
    static const bool want_stamps = getenv("HIPSPARK_FINISH_STAMPS") && getenv("HIPSPARK_FINISH_STAMPS")[0] == '1';
    A.stamps = want_stamps ? (int64_t*)((char*)scratch + (((size_t)cap * 8 * (size_t)(fin->n_fold + 2) + 64 + 7) & ~(size_t)7)) : nullptr;
    // result image size, and the working set that may move into LDS behind the merge core's block
    int64_t image_bytes = 16;
    for (int o = 0; o < fin->n_out; ++o) {
        const hs_finish_out& d = fin->outs[o];
        const int width = d.src == 0 ? key_bytes : (d.kind == HS_I64 ? 8 : 4);
        const int64_t end = d.offset + (int64_t)cap * width;
        if (end > image_bytes) image_bytes = end;
    }
    image_bytes = (image_bytes + 15) & ~(int64_t)15;
    A.image_bytes = image_bytes;
    size_t total_lds = (lds + 15) & ~(size_t)15;
    A.lds_stack = (int64_t)total_lds;  // projection stack: (HS_MAX_STACK + 1) cells x one wave
    total_lds += (size_t)(HS_MAX_STACK + 1) * HS_WAVE * 8;
    const size_t slab_bytes = ((size_t)world * (size_t)desc->stride + 15) & ~(size_t)15;
    const size_t work_bytes = ((size_t)cap * 8 * (size_t)(fin->n_fold + 2) + 15) & ~(size_t)15;
    const bool small = desc->stride % 8 == 0 && total_lds + slab_bytes + work_bytes + (size_t)image_bytes <= HS_MERGE_LDS_MAX;
    if (small) {
        A.lds_slabs = (int64_t)total_lds;
        A.lds_work = A.lds_slabs + (int64_t)slab_bytes;
        A.lds_image = A.lds_work + (int64_t)work_bytes;
        total_lds += slab_bytes + work_bytes + (size_t)image_bytes;
    }
    if (total_lds > HS_MERGE_LDS_MAX) {
        hs_set_error("hs_agg_finish: %lld partial rows x %d aggregates need %zu B LDS (> %zu)", (long long)n_rows,
                     fin->n_fold, total_lds, HS_MERGE_LDS_MAX);
        return HS_E_LIMIT;
    }
    static unsigned long long attr_set = 0;
    if (hs_first_on_device(attr_set)) {
        hipFuncSetAttribute((const void*)k_agg_finish<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HS_MERGE_LDS_MAX);
        hipFuncSetAttribute((const void*)k_agg_finish<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HS_MERGE_LDS_MAX);
    }
    const dim3 block(merge_block(n_rows, cap, A.n_order));
    if (small) hipLaunchKernelGGL(k_agg_finish<true>, dim3(1), block, total_lds, (hipStream_t)stream, A);
    else hipLaunchKernelGGL(k_agg_finish<false>, dim3(1), block, total_lds, (hipStream_t)stream, A);
    if (hipGetLastError() != hipSuccess) {
        hs_set_error("hs_agg_finish: kernel launch failed");
        return HS_E_LAUNCH;
    }
    return HS_OK;
}

extern "C" int hs_host_device_pointer(void* host_ptr, void** device_ptr) {
    if (!host_ptr || !device_ptr) {
        hs_set_error("hs_host_device_pointer: null argument");
        return HS_E_ARG;
    }
    *device_ptr = nullptr;
    if (hipHostGetDevicePointer(device_ptr, host_ptr, 0) != hipSuccess || !*device_ptr) {
        (void)hipGetLastError();
        hs_set_error("hs_host_device_pointer: the allocation is not mapped into the device");
        return HS_E_LAUNCH;
    }
    return HS_OK;
}

// ---- shared-dictionary tier: geometry and launch ------------------------------------------------------------------
int hs_jit_launch_agg_shared(const AggMainArgs* args, bool hashed, unsigned grid, unsigned block, size_t lds_bytes,
                             hipStream_t stream);

static constexpr int HS_SHARED_WG = 1024;

extern "C" int hs_agg_shared_geom(const int64_t* host_unit_rows, int64_t n_units, int32_t n_acc, int32_t group_cap,
                                  hs_agg_geom* out) {
    if (!host_unit_rows || !out || n_units < 0 || n_acc < 0 || n_acc > HS_MAX_ACC || group_cap < 1 ||
        (group_cap & (group_cap - 1))) {
        hs_set_error("hs_agg_shared_geom: bad arguments");
        return HS_E_ARG;
    }
    // LDS table: up to 4x the requested capacity (a quarter-full open-addressing table answers in ~1 probe; at 80 %
    // the same kernel ran 3x slower), as far as 128 KiB allow; then accumulator replicas per slot (a lane uses
    // replica lane % R) with what is left, at most one per lane of a wave - with few groups the lanes of a wave
    // would otherwise serialise on the same LDS words
    const size_t cell_bytes = (size_t)(n_acc > 0 ? n_acc : 1) * 8;
    int slots = group_cap;
    while (slots < 4 * group_cap && (size_t)(slots * 2) * (16 + cell_bytes) <= 128 * 1024 && slots < 8192) slots *= 2;
    int replicas = 1;
    while (replicas < HS_WAVE && (size_t)slots * 16 + (size_t)slots * cell_bytes * (size_t)(replicas * 2) <= 128 * 1024)
        replicas *= 2;
    const size_t lds = (size_t)slots * 16 + (size_t)slots * cell_bytes * (size_t)replicas;
    if (lds > HS_LDS_HARD || group_cap > 8192) {
        hs_set_error("hs_agg_shared_geom: a table of %d groups x %d aggregates needs %zu B LDS (> %zu)", group_cap, n_acc,
                     lds, HS_LDS_HARD);
        return HS_E_LIMIT;
    }
    group_cap = slots;
    int64_t total = 0;
    for (int64_t u = 0; u < n_units; ++u) {
        const int64_t r = host_unit_rows[u + 1] - host_unit_rows[u];
        if (r < 0) {
            hs_set_error("hs_agg_shared_geom: unit_rows not ascending at %lld", (long long)u);
            return HS_E_ARG;
        }
        total += r;
    }
    // Two rounds of workgroups even out the tail - but never one workgroup more than the rounds hold: the chunks
    // are equally long, so 515 of them on 256 CUs (one resident workgroup each) take three rounds, not two
    // (measured: 1.13 ms instead of 0.76).  Chunks do not span units, hence the re-count per candidate length.
    const int64_t step = (int64_t)HS_SHARED_WG * HS_V;
    const int64_t resident = lds * 2 + 2048 <= 160 * 1024 ? 2 : 1;  // 1024-lane workgroups: at most two per CU
    const int64_t target = 256 * resident * 2;
    int64_t chunk = ((total + target - 1) / target + step - 1) / step * step;
    if (chunk < 16 * step) chunk = 16 * step;
    if (chunk > 0x40000000) chunk = 0x40000000 / step * step;
    int64_t n_chunks = 0;
    for (int attempt = 0; attempt < 64; ++attempt) {
        n_chunks = 0;
        for (int64_t u = 0; u < n_units; ++u) {
            const int64_t anchor = host_unit_rows[u] & ~(int64_t)(HS_V - 1);
            const int64_t span = host_unit_rows[u + 1] - anchor;
            n_chunks += span > 0 ? (span + chunk - 1) / chunk : 0;
        }
        if (n_chunks <= target || n_units >= target || chunk >= 0x40000000 / step * step) break;
        chunk += step * (1 + chunk / step / 32);
    }
    int32_t unit_cap = group_cap * 2;  // a unit sees at least the groups of any of its chunks: half-full at worst
    out->group_cap = group_cap;
    out->chunk_rows = (int32_t)chunk;
    out->wg_threads = HS_SHARED_WG;
    out->pad = unit_cap;
    out->n_chunks = n_chunks;
    out->lds_bytes = lds;
    out->ws_bytes = (size_t)n_units * (size_t)unit_cap * 8 + 256;  // the unit tables' key words
    return HS_OK;
}

static int agg_shared_impl(void* stream, const hs_col* cols, int32_t n_cols, int32_t key_col, int32_t unit_col,
                           const hs_program* prog, const hs_agg_spec* spec, const hs_chunk* chunks, int64_t n_units,
                           const hs_agg_geom* geom, int64_t* out_rep, uint64_t* out_acc, int32_t* out_ngroups, void* ws,
                           uint32_t* flags, void* ev_begin, void* ev_end, const hs_join8* join = nullptr,
                           uint64_t* unit_keys = nullptr);

extern "C" int hs_agg_shared(void* stream, const hs_col* cols, int32_t n_cols, int32_t key_col, const hs_program* prog,
                             const hs_agg_spec* spec, const hs_chunk* chunks, int64_t n_units, const hs_agg_geom* geom,
                             int64_t* out_rep, uint64_t* out_acc, int32_t* out_ngroups, void* ws, uint32_t* flags,
                             void* ev_begin, void* ev_end) {
    return agg_shared_impl(stream, cols, n_cols, key_col, -1, prog, spec, chunks, n_units, geom, out_rep, out_acc,
                           out_ngroups, ws, flags, ev_begin, ev_end);
}

extern "C" int hs_agg_shared_units(void* stream, const hs_col* cols, int32_t n_cols, int32_t key_col, int32_t unit_col,
                                   int32_t n_unit_tables, const hs_program* prog, const hs_agg_spec* spec,
                                   const hs_chunk* chunks, const hs_agg_geom* geom, int64_t* out_rep, uint64_t* out_acc,
                                   int32_t* out_ngroups, void* ws, uint32_t* flags, void* ev_begin, void* ev_end) {
    if (!cols || unit_col < 0 || unit_col >= n_cols || unit_col >= HS_FUSED_COLS || cols[unit_col].kind != HS_U8 ||
        n_unit_tables < 1 || n_unit_tables > 127 || key_col < 0 || key_col >= n_cols) {
        hs_set_error("hs_agg_shared_units: the unit column must be a HS_U8 column in a preloaded slot, 1..127 units");
        return HS_E_ARG;
    }
    const hs_col& kc = cols[key_col];
    const bool key_ok = kc.kind == HS_I32 || kc.kind == HS_U8 || (kc.kind == HS_STR && kc.fixed_len >= 1 && kc.fixed_len <= 6);
    if (!key_ok) {
        hs_set_error("hs_agg_shared_units: the key must fit 56 bits (INTEGER, or a string of fixed length <= 6)");
        return HS_E_LIMIT;
    }
    return agg_shared_impl(stream, cols, n_cols, key_col, unit_col, prog, spec, chunks, n_unit_tables, geom, out_rep,
                           out_acc, out_ngroups, ws, flags, ev_begin, ev_end);
}

static int agg_shared_impl(void* stream, const hs_col* cols, int32_t n_cols, int32_t key_col, int32_t unit_col,
                           const hs_program* prog, const hs_agg_spec* spec, const hs_chunk* chunks, int64_t n_units,
                           const hs_agg_geom* geom, int64_t* out_rep, uint64_t* out_acc, int32_t* out_ngroups, void* ws,
                           uint32_t* flags, void* ev_begin, void* ev_end, const hs_join8* join, uint64_t* unit_keys) {
    // join != NULL: the fused probe (hs_agg_shared_join8) - unit tables stay raw (no rounding pass, no group counts)
    if (!cols || !prog || !spec || !chunks || !geom || !out_rep || !out_acc || (!out_ngroups && !join) || !ws || !flags ||
        key_col < 0 || key_col >= n_cols) {
        hs_set_error("hs_agg_shared: null or out-of-range argument");
        return HS_E_ARG;
    }
    if (prog->n_ins > HS_MAX_INS || prog->n_lit > HS_MAX_LIT) {
        hs_set_error("hs_agg_shared: program too long");
        return HS_E_LIMIT;
    }
    // slots of a unit's table: with row-range units at least the LDS table's (a unit sees every group of its chunks);
    // with computed units the caller sizes it for the groups ONE unit holds (an overflow raises HS_FLAG_DICT_FULL)
    if (geom->wg_threads != HS_SHARED_WG || (geom->pad & (geom->pad - 1)) ||
        (unit_col < 0 ? geom->pad < geom->group_cap : geom->pad < 16)) {
        hs_set_error("hs_agg_shared: geometry not made by hs_agg_shared_geom");
        return HS_E_ARG;
    }
    if (n_units == 0 || (geom->n_chunks == 0 && !join)) return HS_OK;  // (a rank without probe rows still clears its tables)
    AggMainArgs A;
    int rc = fill_cols(A.cols, cols, n_cols);
    if (rc) return rc;
    for (int i = HS_FUSED_COLS; i < n_cols; ++i) {
        if (cols[i].kind != HS_STR || i == key_col) {
            hs_set_error("hs_agg_shared: more than %d numeric column slots", HS_FUSED_COLS);
            return HS_E_LIMIT;
        }
    }
    for (int i = 0; i < n_cols; ++i) {
        if ((cols[i].kind == HS_JOIN8_CODE || cols[i].kind == HS_JOIN8_UNIT) && !join) {
            hs_set_error("hs_agg_shared: HS_JOIN8_* columns belong to hs_agg_shared_join8");
            return HS_E_ARG;
        }
    }
    const int depth = program_depth(prog);
    if (depth > HS_MAX_STACK) {
        hs_set_error("hs_agg_shared: expression stack depth %d > %d", depth, HS_MAX_STACK);
        return HS_E_LIMIT;
    }
    if (geom->n_chunks > 0x7fffffffll) {
        hs_set_error("hs_agg_shared: too many chunks");
        return HS_E_LIMIT;
    }
    A.prog = *prog;
    A.spec = *spec;
    A.key_col = key_col;
    A.group_cap = geom->group_cap;
    A.chunk_rows = geom->chunk_rows;
    A.pad = geom->pad;
    A.chunks = chunks;
    A.unit_chunk0 = nullptr;
    A.n_units = n_units;
    A.part_keys = unit_keys ? unit_keys : (uint64_t*)ws;
    A.part_rep = out_rep;
    A.part_acc = out_acc;
    A.flags = flags;
    A.unit_arrivals = nullptr;  // this tier merges chunk tables with global atomics, there is no unit combine
    memset(&A.unit, 0, sizeof(A.unit));
    A.unit_col = unit_col;  // -1: units are the chunks' row ranges; else n_units = number of unit tables
    A.pad3 = 0;
    // computed units: per-chunk cells behind the unit tables' key words in `ws` (hs_agg_shared_units documents the size);
    // with caller-owned key words (hs_agg_shared_join8) `ws` holds the per-chunk cells only
    A.chunk_acc = unit_col < 0 ? nullptr : unit_keys ? (uint64_t*)ws
                : (uint64_t*)((char*)ws + (((size_t)n_units * (size_t)geom->pad * 8 + 256 + 15) & ~(size_t)15));
    if (join) A.join = *join;
    else memset(&A.join, 0, sizeof(A.join));
    A.stamps = nullptr;
    {  // the replica count the geometry sized the LDS block for
        const size_t per_replica = (size_t)geom->group_cap * (size_t)(spec->n_acc > 0 ? spec->n_acc : 1) * 8;
        const size_t r = (geom->lds_bytes - (size_t)geom->group_cap * 16) / per_replica;
        if (r < 1 || r > HS_WAVE || (r & (r - 1)) || (size_t)geom->group_cap * 16 + per_replica * r != geom->lds_bytes) {
            hs_set_error("hs_agg_shared: geometry not made by hs_agg_shared_geom");
            return HS_E_ARG;
        }
        A.replicas = (int32_t)r;
        // the table counts as overflowed once it is half full (lockstep probing: see SharedCtx) - unless it is the
        // largest one LDS holds: past that only the HBM tier is left, and a crowded LDS table still beats it
        const size_t slot_bytes = 16 + (size_t)(spec->n_acc > 0 ? spec->n_acc : 1) * 8;
        const bool can_grow = (size_t)geom->group_cap * 2 * slot_bytes <= 128 * 1024 && geom->group_cap * 2 <= 8192;
        A.pad2 = can_grow ? geom->group_cap / 2 : geom->group_cap;
    }
    hipStream_t s = (hipStream_t)stream;
    SharedInitArgs I;
    I.keys = A.part_keys;
    I.reps = out_rep;
    I.acc = out_acc;
    I.n_slots = n_units * (int64_t)geom->pad;
    I.spec = *spec;
    int64_t init_blocks = (I.n_slots + 255) / 256;
    if (init_blocks > 4096) init_blocks = 4096;
    hipLaunchKernelGGL(k_agg_shared_init, dim3((unsigned)init_blocks), dim3(256), 0, s, I);
    if (geom->n_chunks == 0) {  // join mode, no rows on this rank: empty tables are its share
        if (ev_begin) hs_event_record((hipEvent_t)ev_begin, s);
        if (ev_end) hs_event_record((hipEvent_t)ev_end, s);
        return hipGetLastError() == hipSuccess ? HS_OK : HS_E_LAUNCH;
    }
    const bool hashed = !hs_col_packs(cols[key_col]);
    static unsigned long long attrs_set = 0;
    if (hs_first_on_device(attrs_set)) {
        allow_big_lds(k_agg_shared<true, 8>);
        allow_big_lds(k_agg_shared<false, 4>);
        allow_big_lds(k_agg_shared<false, 8>);
    }
    dim3 grid((unsigned)geom->n_chunks), block((unsigned)HS_SHARED_WG);
    if (ev_begin) hs_event_record((hipEvent_t)ev_begin, s);
    const int jit_rc = hs_jit_launch_agg_shared(&A, hashed, grid.x, block.x, geom->lds_bytes, s);
    if (jit_rc == HS_OK) {
        // launched the program compiled for exactly this bytecode
    } else if (join) {
        // the interpreter kernels do not know the virtual columns; the init launch above is harmless
        hs_set_error("hs_agg_shared_join8: needs the run-time compiler (hiprtc): %s", hs_jit_last_log());
        return HS_E_LIMIT;
    } else if (hashed) {
        hipLaunchKernelGGL((k_agg_shared<true, 8>), grid, dim3(hs_shared_interp_wg(8)), geom->lds_bytes, s, A);
    } else if (depth <= 4) {
        hipLaunchKernelGGL((k_agg_shared<false, 4>), grid, dim3(hs_shared_interp_wg(4)), geom->lds_bytes, s, A);
    } else {
        hipLaunchKernelGGL((k_agg_shared<false, 8>), grid, dim3(hs_shared_interp_wg(8)), geom->lds_bytes, s, A);
    }
    if (A.chunk_acc && spec->n_acc > 0) {
        SharedFoldArgs G;
        G.chunk_acc = A.chunk_acc;
        G.reps = out_rep;
        G.acc = out_acc;
        G.n_chunks = geom->n_chunks;
        G.n_units = (int32_t)n_units;
        G.unit_cap = geom->pad;
        G.spec = *spec;
        const int64_t tasks = n_units * (int64_t)geom->pad * (spec->n_acc > 0 ? spec->n_acc : 1);
        hipLaunchKernelGGL(k_agg_shared_fold_chunks, dim3((unsigned)((tasks + 3) / 4)), dim3(256), 0, s, G);
    }
    if (ev_end) hs_event_record((hipEvent_t)ev_end, s);
    if (join) {
        if (hipGetLastError() != hipSuccess) {
            hs_set_error("hs_agg_shared_join8: kernel launch failed");
            return HS_E_LAUNCH;
        }
        return HS_OK;  // raw tables: hs_agg_units_merge (N ranks) / hs_agg_units_to_slab round and emit them
    }
    SharedFinishArgs F;
    F.reps = out_rep;
    F.acc = out_acc;
    F.ngroups = out_ngroups;
    F.unit_cap = geom->pad;
    F.pad = 0;
    F.spec = *spec;
    F.flags = flags;
    hipLaunchKernelGGL(k_agg_shared_finish, dim3((unsigned)n_units), dim3(256), 0, s, F);
    if (hipGetLastError() != hipSuccess) {
        hs_set_error("hs_agg_shared: kernel launch failed");
        return HS_E_LAUNCH;
    }
    return HS_OK;
}

// ---- round 3: the join's probe inside the scan; raw unit tables -> [merge over ranks] -> exchange slab ----------------
extern "C" const char* hs_jit_last_log(void);

extern "C" int hs_agg_shared_join8(void* stream, const hs_col* cols, int32_t n_cols, int32_t key_col, int32_t unit_col,
                                   const hs_join8* join, int32_t n_unit_tables, const hs_program* prog, const hs_agg_spec* spec,
                                   const hs_chunk* chunks, const hs_agg_geom* geom, int64_t* out_rep, uint64_t* unit_keys,
                                   uint64_t* unit_acc, void* ws, uint32_t* flags, void* ev_begin, void* ev_end) {
    if (!cols || !join || !join->table || join->slots < 1 || join->n_parts < 1 || join->n_parts > 127 || !unit_keys ||
        unit_col < 0 || unit_col >= n_cols || unit_col >= HS_FUSED_COLS || cols[unit_col].kind != HS_JOIN8_UNIT ||
        n_unit_tables != join->n_parts || key_col < 0 || key_col >= n_cols) {
        hs_set_error("hs_agg_shared_join8: the unit column must be a HS_JOIN8_UNIT column in a preloaded slot, one unit "
                     "table per shuffle partition (1..127)");
        return HS_E_ARG;
    }
    const void* probe_keys = cols[unit_col].data;
    for (int i = 0; i < n_cols; ++i) {
        const bool virt = cols[i].kind == HS_JOIN8_CODE || cols[i].kind == HS_JOIN8_UNIT;
        // (a rank without probe rows - geom->n_chunks 0 - has no key column to show: it only clears its tables)
        if (virt && (i >= HS_FUSED_COLS || cols[i].data != probe_keys ||
                     (geom && geom->n_chunks > 0 && (!probe_keys || ((uintptr_t)probe_keys & 15))))) {
            hs_set_error("hs_agg_shared_join8: virtual columns sit in preloaded slots and share one 16-byte aligned key column");
            return HS_E_ARG;
        }
    }
    const hs_col& kc = cols[key_col];
    const bool key_ok = kc.kind == HS_I32 || kc.kind == HS_U8 || kc.kind == HS_JOIN8_CODE ||
                        (kc.kind == HS_STR && kc.fixed_len >= 1 && kc.fixed_len <= 6);
    if (!key_ok) {
        hs_set_error("hs_agg_shared_join8: the key must fit 56 bits (INTEGER, the table byte, or a string of fixed length <= 6)");
        return HS_E_LIMIT;
    }
    return agg_shared_impl(stream, cols, n_cols, key_col, unit_col, prog, spec, chunks, n_unit_tables, geom, out_rep,
                           unit_acc, nullptr, ws, flags, ev_begin, ev_end, join, unit_keys);
}

// One workgroup per unit.  Rank after rank: every occupied slot of the rank's table is upserted into the unit's
// output table (a rank holds a key at most once, so no two lanes meet in one output slot within a step) and its cells
// are folded in - additions in rank order.  Output accesses are agent-scope atomics (they bypass the CU's vector
// cache: lanes of LATER steps read what other lanes of this workgroup wrote).
struct UnitsMergeArgs {
    const uint8_t* gathered;
    int64_t stride;  // bytes per rank
    int32_t world, n_units, unit_cap, pad;
    hs_agg_spec spec;
    uint64_t* out_keys;
    uint64_t* out_acc;
    uint32_t* flags;
};
__global__ void __launch_bounds__(256) k_agg_units_merge(const UnitsMergeArgs A) {
    const int NA = A.spec.n_acc, UC = A.unit_cap, tid = threadIdx.x, nthr = blockDim.x;
    const int64_t u = blockIdx.x, slots = (int64_t)A.n_units * UC;
    uint64_t* okeys = A.out_keys + u * UC;
    uint64_t* oacc = A.out_acc + u * (int64_t)UC * NA;
    for (int sl = tid; sl < UC; sl += nthr) {
        __hip_atomic_store(&okeys[sl], HS_EMPTY_KEY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int a = 0; a < NA; ++a)
            __hip_atomic_store(&oacc[(int64_t)sl * NA + a], hs_acc_identity(A.spec.op[a], A.spec.is_int[a] != 0), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    uint32_t err = 0;
    const uint32_t mask = (uint32_t)UC - 1;
    for (int r = 0; r < A.world; ++r) {
        const uint8_t* base = A.gathered + (int64_t)r * A.stride;
        if (u == 0 && tid == 0) err |= *(const uint32_t*)base;  // the rank's status word travels in the header
        const uint64_t* rkeys = (const uint64_t*)(base + 16) + u * UC;
        const uint64_t* racc = (const uint64_t*)(base + 16) + slots + u * (int64_t)UC * NA;
        for (int sl = tid; sl < UC; sl += nthr) {
            const uint64_t k = rkeys[sl];
            if (k == HS_EMPTY_KEY) continue;
            uint32_t h = hs_slot_hash_strong(k) & mask;
            int found = -1;
            for (uint32_t probe = 0; probe <= mask; ++probe) {
                uint64_t cur = __hip_atomic_load(&okeys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == HS_EMPTY_KEY) {
                    cur = atomicCAS((unsigned long long*)&okeys[h], (unsigned long long)HS_EMPTY_KEY, (unsigned long long)k);
                    if (cur == HS_EMPTY_KEY) cur = k;
                }
                if (cur == k) {
                    found = (int)h;
                    break;
                }
                h = (h + 1) & mask;
            }
            if (found < 0) {
                err |= HS_FLAG_DICT_FULL;
                continue;
            }
            for (int a = 0; a < NA; ++a) {
                uint64_t* cell = &oacc[(int64_t)found * NA + a];
                const uint64_t v = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cell, hs_acc_fold(A.spec.op[a], A.spec.is_int[a] != 0, v, racc[(int64_t)sl * NA + a]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();  // the next rank's rows may land in slots this step filled
    }
    if (err) atomicOr(A.flags, err);
}

extern "C" int hs_agg_units_merge(void* stream, const uint8_t* gathered, int32_t world, int32_t n_units, int32_t unit_cap,
                                  const hs_agg_spec* spec, uint64_t* out_keys, uint64_t* out_acc, uint32_t* flags) {
    if (!gathered || !spec || !out_keys || !out_acc || !flags || world < 1 || n_units < 1 || n_units > 127 || unit_cap < 1 ||
        (unit_cap & (unit_cap - 1)) || spec->n_acc < 0 || spec->n_acc > HS_MAX_ACC || ((uintptr_t)gathered & 7)) {
        hs_set_error("hs_agg_units_merge: bad arguments");
        return HS_E_ARG;
    }
    UnitsMergeArgs A;
    A.gathered = gathered;
    A.stride = 16 + (int64_t)n_units * unit_cap * 8 * (1 + spec->n_acc);
    A.world = world;
    A.n_units = n_units;
    A.unit_cap = unit_cap;
    A.pad = 0;
    A.spec = *spec;
    A.out_keys = out_keys;
    A.out_acc = out_acc;
    A.flags = flags;
    hipLaunchKernelGGL(k_agg_units_merge, dim3((unsigned)n_units), dim3(256), 0, (hipStream_t)stream, A);
    if (hipGetLastError() != hipSuccess) {
        hs_set_error("hs_agg_units_merge: kernel launch failed");
        return HS_E_LAUNCH;
    }
    return HS_OK;
}

struct UnitsSlabArgs {
    const uint64_t* keys;
    const uint64_t* acc;
    int32_t n_units, unit_cap;
    hs_agg_spec spec;
    uint8_t* slab;
    hs_slab_desc desc;
    uint32_t* flags;
};
__global__ void __launch_bounds__(256) k_agg_units_to_slab(const UnitsSlabArgs A) {
    const int NA = A.spec.n_acc;
    const int64_t n = (int64_t)A.n_units * A.unit_cap;
    int64_t* order = (int64_t*)(A.slab + A.desc.order_off);
    const int kb = A.desc.key_kind == HS_STR ? A.desc.key_len : (A.desc.key_kind == HS_U8 ? 1 : 4);
    uint32_t err = 0;
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < A.desc.slab_rows; row += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t k = row < n ? A.keys[row] : HS_EMPTY_KEY;
        if (k == HS_EMPTY_KEY) {
            order[row] = -1;
            continue;
        }
        order[row] = row / A.unit_cap;
        uint8_t* kp = A.slab + A.desc.key_off + row * kb;
        for (int b = 0; b < kb; ++b) kp[b] = (uint8_t)(k >> (8 * b));  // packed strings and INTEGER keys: little-endian low bytes
        for (int a = 0; a < NA; ++a) {
            uint64_t cell = A.acc[row * NA + a];
            const bool is_int = A.spec.is_int[a] != 0;
            if (hs_float_identity_left(A.spec.op[a], is_int, cell)) err |= HS_FLAG_TYPE_ASSERT;
            cell = hs_quantise_cell(is_int, cell, err);
            uint8_t* col = A.slab + A.desc.acc_off[a];
            if (A.desc.acc_kind[a] == HS_I32) ((int32_t*)col)[row] = (int32_t)(int64_t)cell;
            else ((float*)col)[row] = (float)hs_u2d(cell);
        }
    }
    if (err) atomicOr(A.flags, err);
}

extern "C" int hs_agg_units_to_slab(void* stream, const uint64_t* unit_keys, const uint64_t* unit_acc, int32_t n_units,
                                    int32_t unit_cap, const hs_agg_spec* spec, uint8_t* slab, const hs_slab_desc* desc,
                                    uint32_t* flags) {
    if (!unit_keys || !unit_acc || !spec || !slab || !desc || !flags || n_units < 1 || unit_cap < 1 ||
        desc->slab_rows < (int64_t)n_units * unit_cap || desc->n_acc != spec->n_acc || spec->n_acc > HS_MAX_ACC) {
        hs_set_error("hs_agg_units_to_slab: bad arguments (the slab must hold n_units * unit_cap rows)");
        return HS_E_ARG;
    }
    const bool key_ok = desc->key_kind == HS_I32 || desc->key_kind == HS_U8 ||
                        (desc->key_kind == HS_STR && (desc->key_len == 1 || desc->key_len == 2 || desc->key_len == 4));
    if (!key_ok) {
        hs_set_error("hs_agg_units_to_slab: key kind %d cannot be rebuilt from a key word", desc->key_kind);
        return HS_E_LIMIT;
    }
    for (int a = 0; a < spec->n_acc; ++a) {
        if (desc->acc_kind[a] != (spec->is_int[a] ? HS_I32 : HS_F32)) {
            hs_set_error("hs_agg_units_to_slab: slab column %d does not hold the aggregate's stored kind", a);
            return HS_E_ARG;
        }
    }
    UnitsSlabArgs A;
    A.keys = unit_keys;
    A.acc = unit_acc;
    A.n_units = n_units;
    A.unit_cap = unit_cap;
    A.spec = *spec;
    A.slab = slab;
    A.desc = *desc;
    A.flags = flags;
    int64_t blocks = (desc->slab_rows + 255) / 256;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(k_agg_units_to_slab, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, A);
    if (hipGetLastError() != hipSuccess) {
        hs_set_error("hs_agg_units_to_slab: kernel launch failed");
        return HS_E_LAUNCH;
    }
    return HS_OK;
}
