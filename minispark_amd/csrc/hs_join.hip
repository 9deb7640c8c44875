// hs_join.hip - round-2 operators for the join / string-key paths of BASELINE configs 4 and 5:
//
//   hs_dict_build / hs_dict_assign   dictionary-encode a STRING column with few distinct values (table open): one
//                                    u8 code per row; LIKE / comparisons / concatenation / GROUP BY then run on codes
//   hs_dict_combine                  codes of a concatenation of dictionary-coded columns (mixed radix)
//   hs_minmax_i32                    key range of a join's build side
//   hs_join_build_unique / hs_join_probe_unique
//                                    primary-key / foreign-key join: ONE 32-bit word per table slot holds the build
//                                    row; direct addressing when the key range is dense (TPC-H order keys: 4 slots
//                                    per key), multiplicative hashing + linear probing otherwise.  The probe leaves
//                                    the probe side's rows in place and emits, per probe row, the matching build row
//                                    and the row's shuffle partition hash(key) % n_parts (= the reference's JoinJob,
//                                    plan.py:99-109) as a per-row UNIT id for the fused partial aggregate
//                                    (hs_agg_shared_units): joined rows are never materialised or re-ordered.
//
// Reference loops replaced: BroadcastHashJoinTask.generate_chunks tasks.py:201-240 (zig-src/src/tasks.zig:70-194),
// WriteToShufflePartitions.write tasks.py:347-375 for the two join inputs, LikeColumn / BinaryOperatorColumn on
// strings sql.py:166-212, 262-266.  All HBM-bound integer / byte work.
#include <stdlib.h>

#include "hs_device.h"

extern thread_local char g_hs_err[256];
void hs_set_error(const char* fmt, ...);

#define HSJ_CHECK_LAUNCH(name)                              \
    if (hipGetLastError() != hipSuccess) {                  \
        hs_set_error(name ": kernel launch failed");        \
        return HS_E_LAUNCH;                                 \
    }

static unsigned hsj_grid(int64_t n, int per_block, int64_t max_blocks = 1 << 16) {
    int64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (unsigned)b;
}

// =====================================================================================================
// Dictionary encoding
// =====================================================================================================
// The table (global memory, L2-resident) is an open-addressing set of the column's distinct strings, a slot = a
// representative row.  Almost every row finds its string present: a read-only probe + one short byte compare.
__device__ __forceinline__ int hsj_dict_find_or_insert(uint64_t* words, int64_t* reps, uint32_t mask, const hs_col& c,
                                                       int64_t row, bool insert, int32_t* count, bool& full) {
    // the slot is claimed by CAS on its representative row; equality is ALWAYS decided on the bytes of that row
    // (immutable input), exactly like the global dictionaries of the join / GROUP BY builds (hs_ops.hip gdict_upsert);
    // words[] keeps the key word of the slot's string for information only
    const uint64_t k = hs_key_at(c, row);
    uint32_t h = (uint32_t)(hs_mix64(k) >> 24) & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        long long rep = __hip_atomic_load(&reps[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (rep < 0) {
            if (!insert) return -1;
            rep = (long long)atomicCAS((unsigned long long*)&reps[h], (unsigned long long)(-1ll), (unsigned long long)row);
            if (rep < 0) {
                words[h] = k;
                if (atomicAdd(count, 1) >= (int)(mask >> 1)) full = true;  // half full: give up (see k_dict_build)
                return (int)h;
            }
        }
        if (hs_rows_equal(c, (int64_t)rep, row)) return (int)h;
        h = (h + 1) & mask;
    }
    full = true;
    return -1;
}

__global__ void __launch_bounds__(256) k_dict_init(uint64_t* words, int64_t* reps, int32_t cap, int32_t* count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) {
        words[i] = HS_EMPTY_KEY;
        reps[i] = -1;
    }
    if (i == 0) *count = 0;
}

__global__ void __launch_bounds__(256) k_dict_build(const hs_col col, int64_t n, int32_t cap, uint64_t* words, int64_t* reps,
                                                    int32_t* count, uint32_t* flags) {
    __shared__ int s_stop;
    bool full = false;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < n; base += (int64_t)gridDim.x * blockDim.x) {
        // far more entries than a code byte can name: the column is not encoded - stop reading it (uniform decision
        // per round).  The threshold is half the table, not 256: under heavy contention (millions of rows, a handful
        // of strings, every workgroup inserting the same few at once) a string can end up in more than one slot, so
        // `count` over-counts; the caller names codes by STRING (duplicate slots share a code) and applies the
        // 256 limit to the distinct strings it reads back.
        if (threadIdx.x == 0) s_stop = __hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > cap / 2;
        __syncthreads();
        if (s_stop) break;
        const int64_t row = base + threadIdx.x;
        if (row < n) hsj_dict_find_or_insert(words, reps, (uint32_t)cap - 1, col, row, true, count, full);
        __syncthreads();
    }
    if (full) atomicOr(flags, HS_FLAG_DICT_FULL);
}

__global__ void __launch_bounds__(256) k_dict_assign(const hs_col col, int64_t n, int32_t cap, uint64_t* words, int64_t* reps,
                                                     const uint8_t* slot_code, uint8_t* out, uint32_t* flags) {
    bool full = false;
    uint32_t err = 0;
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n; row += (int64_t)gridDim.x * blockDim.x) {
        const int s = hsj_dict_find_or_insert(words, reps, (uint32_t)cap - 1, col, row, false, nullptr, full);
        if (s < 0) err |= HS_FLAG_BAD_PROGRAM;  // cannot happen after a complete build
        out[row] = s < 0 ? 0 : slot_code[s];
    }
    if (err) atomicOr(flags, err);
}

extern "C" int hs_dict_build(void* stream, const hs_col* col, int64_t nrows, int32_t cap, uint64_t* slot_words,
                             int64_t* slot_reps, int32_t* count, uint32_t* flags) {
    if (!col || col->kind != HS_STR || !slot_words || !slot_reps || !count || !flags || cap < 512 || (cap & (cap - 1)) ||
        nrows < 0) {
        hs_set_error("hs_dict_build: bad arguments (STRING column, cap = power of two >= 512)");
        return HS_E_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_dict_init, dim3((cap + 255) / 256), dim3(256), 0, s, slot_words, slot_reps, cap, count);
    if (nrows > 0)
        hipLaunchKernelGGL(k_dict_build, dim3(hsj_grid(nrows, 256, 2048)), dim3(256), 0, s, *col, nrows, cap, slot_words,
                           slot_reps, count, flags);
    HSJ_CHECK_LAUNCH("hs_dict_build");
    return HS_OK;
}

extern "C" int hs_dict_assign(void* stream, const hs_col* col, int64_t nrows, int32_t cap, uint64_t* slot_words,
                              int64_t* slot_reps, const uint8_t* slot_code, uint8_t* out_codes, uint32_t* flags) {
    if (!col || col->kind != HS_STR || !slot_words || !slot_reps || !slot_code || !out_codes || !flags || cap < 512 ||
        (cap & (cap - 1)) || nrows < 0) {
        hs_set_error("hs_dict_assign: bad arguments");
        return HS_E_ARG;
    }
    if (nrows > 0)
        hipLaunchKernelGGL(k_dict_assign, dim3(hsj_grid(nrows, 256, 4096)), dim3(256), 0, (hipStream_t)stream, *col, nrows, cap,
                           slot_words, slot_reps, slot_code, out_codes, flags);
    HSJ_CHECK_LAUNCH("hs_dict_assign");
    return HS_OK;
}

// codes of `a + lit + b + ...` over dictionary-coded columns: out = sum_k codes_k * stride_k (mixed radix; the
// caller builds the product dictionary in the same order).  Sixteen rows per lane with 16-byte loads and stores.
struct DictCombineArgs {
    const uint8_t* codes[4];
    int32_t stride[4];
    int32_t n_parts, pad;
};
__global__ void __launch_bounds__(256) k_dict_combine(const DictCombineArgs A, int64_t n, uint8_t* out) {
    const int64_t ngroups = (n + 15) / 16;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t base = g * 16;
        uint32_t acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0;
        for (int k = 0; k < A.n_parts; ++k) {
            const uint4 v = *reinterpret_cast<const uint4*>(A.codes[k] + base);  // buffers carry 64 bytes of slack
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] += ((w[j >> 2] >> (8 * (j & 3))) & 0xffu) * (uint32_t)A.stride[k];
        }
        uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j >> 2] |= (acc[j] & 0xffu) << (8 * (j & 3));
        if (base + 16 <= n) {
            *reinterpret_cast<uint4*>(out + base) = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
            for (int j = 0; base + j < n; ++j) out[base + j] = (uint8_t)(o[j >> 2] >> (8 * (j & 3)));
        }
    }
}
extern "C" int hs_dict_combine(void* stream, int32_t n_parts, const uint8_t* const* codes, const int32_t* strides,
                               int64_t nrows, uint8_t* out_codes) {
    if (n_parts < 1 || n_parts > 4 || !codes || !strides || !out_codes || nrows < 0) {
        hs_set_error("hs_dict_combine: bad arguments (1..4 parts)");
        return HS_E_ARG;
    }
    DictCombineArgs A;
    for (int k = 0; k < 4; ++k) {
        A.codes[k] = k < n_parts ? codes[k] : nullptr;
        A.stride[k] = k < n_parts ? strides[k] : 0;
        if (k < n_parts && (!codes[k] || ((uintptr_t)codes[k] & 15) || strides[k] < 0)) {
            hs_set_error("hs_dict_combine: part %d: null / unaligned codes or negative stride", k);
            return HS_E_ARG;
        }
    }
    if ((uintptr_t)out_codes & 15) {
        hs_set_error("hs_dict_combine: output must be 16-byte aligned");
        return HS_E_ARG;
    }
    A.n_parts = n_parts;
    A.pad = 0;
    if (nrows > 0)
        hipLaunchKernelGGL(k_dict_combine, dim3(hsj_grid((nrows + 15) / 16, 256, 8192)), dim3(256), 0, (hipStream_t)stream, A,
                           nrows, out_codes);
    HSJ_CHECK_LAUNCH("hs_dict_combine");
    return HS_OK;
}

// =====================================================================================================
// Unique-key (PK - FK) join on INTEGER keys
// =====================================================================================================
__global__ void k_minmax_i32_init(int32_t* minmax) {
    minmax[0] = 2147483647;
    minmax[1] = (-2147483647 - 1);
}
__global__ void __launch_bounds__(256) k_minmax_i32(const int32_t* v, int64_t n, int32_t* minmax) {
    // 16-byte loads (buffers carry slack past n), wave shuffle reduce, LDS across the four waves, ONE atomic pair per
    // workgroup: per-wave atomics on the two result words serialised to ~200 us for 15 M keys
    __shared__ int32_t s_lo[4], s_hi[4];
    int32_t lo = 2147483647, hi = (-2147483647 - 1);
    const int64_t nquads = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquads; q += (int64_t)gridDim.x * blockDim.x) {
        const int4 x = *reinterpret_cast<const int4*>(v + q * 4);
        const int32_t e[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (q * 4 + j < n) {
                lo = e[j] < lo ? e[j] : lo;
                hi = e[j] > hi ? e[j] : hi;
            }
        }
    }
    for (int d = HS_WAVE / 2; d >= 1; d >>= 1) {
        const int32_t l2 = __shfl_down(lo, d, HS_WAVE), h2 = __shfl_down(hi, d, HS_WAVE);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & (HS_WAVE - 1)) == 0) {
        s_lo[threadIdx.x / HS_WAVE] = lo;
        s_hi[threadIdx.x / HS_WAVE] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            lo = s_lo[w] < lo ? s_lo[w] : lo;
            hi = s_hi[w] > hi ? s_hi[w] : hi;
        }
        atomicMin(&minmax[0], lo);
        atomicMax(&minmax[1], hi);
    }
}
extern "C" int hs_minmax_i32(void* stream, const int32_t* values, int64_t n, int32_t* minmax) {
    if (!values || !minmax || n < 0) {
        hs_set_error("hs_minmax_i32: bad arguments");
        return HS_E_ARG;
    }
    if ((uintptr_t)values & 15) {
        hs_set_error("hs_minmax_i32: values must be 16-byte aligned");
        return HS_E_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_minmax_i32_init, dim3(1), dim3(1), 0, s, minmax);
    if (n > 0) hipLaunchKernelGGL(k_minmax_i32, dim3(hsj_grid(n, 256 * 16, 1024)), dim3(256), 0, s, values, n, minmax);
    HSJ_CHECK_LAUNCH("hs_minmax_i32");
    return HS_OK;
}

static constexpr uint32_t HSJ_EMPTY = 0xffffffffu;

__device__ __forceinline__ uint64_t hsj_hash_slot(int32_t key, uint64_t mask) {
    return (hs_mix64((uint64_t)(uint32_t)key) >> 7) & mask;
}

struct JoinUniqueArgs {
    const int32_t* keys;   // build or probe keys
    int64_t n;
    uint32_t* table;       // [slots] build row per slot, HSJ_EMPTY = none
    const int32_t* build_keys;  // hashed mode: the build side's key column (the table stores rows, keys live here)
    int64_t slots;         // direct: key range; hashed: power of two
    int32_t key_min;       // direct mode: slot = key - key_min
    int32_t direct;
    uint32_t* flags;
};

__global__ void __launch_bounds__(256) k_fill_u32(uint32_t* p, int64_t n, uint32_t v) {
    // sixteen bytes per lane
    const int64_t n4 = n / 4;
    uint4* p4 = reinterpret_cast<uint4*>(p);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        p4[i] = make_uint4(v, v, v, v);
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// direct addressing: plain stores (a duplicate key makes two rows race for one slot: one of them wins) ...
__global__ void __launch_bounds__(256) k_join_scatter_direct(const JoinUniqueArgs A) {
    uint32_t err = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * blockDim.x) {
        // key_min / slots come from the caller (a cached column range): a key outside them must not become a store
        const int64_t off = (int64_t)A.keys[i] - (int64_t)A.key_min;
        if ((uint64_t)off >= (uint64_t)A.slots) {
            err |= HS_FLAG_BAD_PROGRAM;
            continue;
        }
        A.table[off] = (uint32_t)i;
    }
    if (err) atomicOr(A.flags, err);
}
// ... and a streaming pass over the table counts the occupied slots: fewer than build rows = two rows met in one
// slot = duplicate keys (sequential 16-byte reads of the table instead of one random read per build row)
__global__ void __launch_bounds__(256) k_join_count_occupied(const uint32_t* table, int64_t slots, unsigned long long* occupied) {
    __shared__ unsigned long long s_part[4];
    unsigned long long mine = 0;
    const int64_t n4 = slots / 4;
    const uint4* t4 = reinterpret_cast<const uint4*>(table);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 x = t4[i];
        mine += (x.x != HSJ_EMPTY) + (x.y != HSJ_EMPTY) + (x.z != HSJ_EMPTY) + (x.w != HSJ_EMPTY);
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (int64_t)gridDim.x * blockDim.x)
        mine += table[i] != HSJ_EMPTY;
    for (int d = HS_WAVE / 2; d >= 1; d >>= 1) {
        const uint32_t lo = __shfl_down((uint32_t)mine, d, HS_WAVE), hi = __shfl_down((uint32_t)(mine >> 32), d, HS_WAVE);
        mine += ((unsigned long long)hi << 32) | lo;
    }
    if ((threadIdx.x & (HS_WAVE - 1)) == 0) s_part[threadIdx.x / HS_WAVE] = mine;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(occupied, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}
__global__ void k_join_check_occupied(const unsigned long long* occupied, int64_t n_build, uint32_t* flags) {
    if (*occupied != (unsigned long long)n_build) atomicOr(flags, HS_FLAG_JOIN_DUP);
}
// hashed: claim a slot with a 32-bit CAS; a slot whose row carries the same key is a duplicate
__global__ void __launch_bounds__(256) k_join_insert_hashed(const JoinUniqueArgs A) {
    const uint64_t mask = (uint64_t)A.slots - 1;
    uint32_t err = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t key = A.keys[i];
        uint64_t h = hsj_hash_slot(key, mask);
        bool placed = false;
        for (uint64_t probe = 0; probe <= mask && !placed; ++probe) {
            uint32_t cur = __hip_atomic_load(&A.table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == HSJ_EMPTY) {
                cur = atomicCAS(&A.table[h], HSJ_EMPTY, (uint32_t)i);
                if (cur == HSJ_EMPTY) {
                    placed = true;
                    break;
                }
            }
            if (A.keys[cur] == key) {
                err |= HS_FLAG_JOIN_DUP;
                placed = true;
                break;
            }
            h = (h + 1) & mask;
        }
        if (!placed) err |= HS_FLAG_DICT_FULL;
    }
    if (err) atomicOr(A.flags, err);
}

extern "C" int hs_join_build_unique(void* stream, const int32_t* build_keys, int64_t n_build, int32_t key_min,
                                    int64_t slots, int32_t direct, uint32_t* table, uint32_t* flags) {
    // direct mode keeps its occupied-slot counter in the two words behind the table (the caller allocates slots + 4)
    if (!build_keys || !table || !flags || n_build < 0 || n_build >= 0xffffffffll || slots < 1 ||
        (!direct && (slots & (slots - 1)))) {
        hs_set_error("hs_join_build_unique: bad arguments (hashed tables need a power-of-two slot count)");
        return HS_E_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    JoinUniqueArgs A{build_keys, n_build, table, build_keys, slots, key_min, direct, flags};
    hipLaunchKernelGGL(k_fill_u32, dim3(hsj_grid(slots / 4 + 1, 256, 8192)), dim3(256), 0, s, table, slots, HSJ_EMPTY);
    if (n_build > 0) {
        const unsigned grid = hsj_grid(n_build, 256 * 4, 8192);
        if (direct) {
            unsigned long long* occupied = reinterpret_cast<unsigned long long*>(table + ((slots + 1) & ~(int64_t)1));
            hipLaunchKernelGGL(k_join_scatter_direct, dim3(grid), dim3(256), 0, s, A);
            hs_memset_async(occupied, 0, 8, s);
            hipLaunchKernelGGL(k_join_count_occupied, dim3(hsj_grid(slots / 4 + 1, 256 * 4, 4096)), dim3(256), 0, s, table, slots, occupied);
            hipLaunchKernelGGL(k_join_check_occupied, dim3(1), dim3(1), 0, s, occupied, n_build, flags);
        } else {
            hipLaunchKernelGGL(k_join_insert_hashed, dim3(grid), dim3(256), 0, s, A);
        }
    }
    HSJ_CHECK_LAUNCH("hs_join_build_unique");
    return HS_OK;
}

// Probe: four keys per lane (one 16-byte load), the four table reads in flight together.
struct JoinProbeUniqueArgs {
    JoinUniqueArgs t;       // t.keys = probe keys, t.n = probe rows
    int32_t n_parts, pad;
    int64_t* out_row;       // [n] matching build row (0 for a probe row without a match)
    uint8_t* out_unit;      // [n] hash(key) % n_parts, 0xff = no match
    const uint8_t* payload; // optional: a 1-byte column of the BUILD side (dictionary codes), gathered on the way
    uint8_t* out_payload;   // [n]
};
__device__ __forceinline__ uint32_t hsj_lookup(const JoinUniqueArgs& T, int32_t key) {
    if (T.direct) {
        const int64_t off = (int64_t)key - (int64_t)T.key_min;
        return off >= 0 && off < T.slots ? T.table[off] : HSJ_EMPTY;
    }
    const uint64_t mask = (uint64_t)T.slots - 1;
    uint64_t h = hsj_hash_slot(key, mask);
    for (uint64_t probe = 0; probe <= mask; ++probe) {
        const uint32_t cur = T.table[h];
        if (cur == HSJ_EMPTY) return HSJ_EMPTY;
        if (T.build_keys[cur] == key) return cur;
        h = (h + 1) & mask;
    }
    return HSJ_EMPTY;
}
__global__ void __launch_bounds__(256) k_join_probe_unique(const JoinProbeUniqueArgs A) {
    const int64_t n = A.t.n, nquads = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquads; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t base = q * 4;
        const int4 kv = *reinterpret_cast<const int4*>(A.t.keys + base);  // buffers carry slack past the last row
        const int32_t k[4] = {kv.x, kv.y, kv.z, kv.w};
        // foreign keys usually arrive clustered (TPC-H: the lineitems of an order are adjacent): a row whose key
        // equals its predecessor's re-uses that lookup, so a lane's quad costs one table + one payload access, not four
        uint32_t r[4];
        r[0] = base < n ? hsj_lookup(A.t, k[0]) : HSJ_EMPTY;
#pragma unroll
        for (int j = 1; j < 4; ++j) r[j] = base + j >= n ? HSJ_EMPTY : (k[j] == k[j - 1] ? r[j - 1] : hsj_lookup(A.t, k[j]));
        uint8_t pay[4] = {0, 0, 0, 0};
        if (A.payload) {
            pay[0] = r[0] != HSJ_EMPTY ? A.payload[r[0]] : 0;
#pragma unroll
            for (int j = 1; j < 4; ++j) pay[j] = r[j] == r[j - 1] ? pay[j - 1] : (r[j] != HSJ_EMPTY ? A.payload[r[j]] : 0);
        }
        uint32_t units = 0, pays = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t u = r[j] != HSJ_EMPTY ? hs_py_partition(k[j], A.n_parts) : 0xffu;
            units |= u << (8 * j);
            pays |= (uint32_t)pay[j] << (8 * j);
        }
        if (base + 4 <= n) {
            if (A.out_row) {
                int64_t* o = A.out_row + base;
                reinterpret_cast<longlong2*>(o)[0] = make_longlong2(r[0] != HSJ_EMPTY ? r[0] : 0, r[1] != HSJ_EMPTY ? r[1] : 0);
                reinterpret_cast<longlong2*>(o)[1] = make_longlong2(r[2] != HSJ_EMPTY ? r[2] : 0, r[3] != HSJ_EMPTY ? r[3] : 0);
            }
            *reinterpret_cast<uint32_t*>(A.out_unit + base) = units;
            if (A.out_payload) *reinterpret_cast<uint32_t*>(A.out_payload + base) = pays;
        } else {
            for (int j = 0; base + j < n; ++j) {
                if (A.out_row) A.out_row[base + j] = r[j] != HSJ_EMPTY ? r[j] : 0;
                A.out_unit[base + j] = (uint8_t)(units >> (8 * j));
                if (A.out_payload) A.out_payload[base + j] = pay[j];
            }
        }
    }
}

extern "C" int hs_join_probe_unique(void* stream, const int32_t* probe_keys, int64_t n_probe, const int32_t* build_keys,
                                    int32_t key_min, int64_t slots, int32_t direct, const uint32_t* table, int32_t n_parts,
                                    int64_t* out_build_row, uint8_t* out_unit, const uint8_t* build_payload,
                                    uint8_t* out_payload) {
    if (!probe_keys || !build_keys || !table || !out_unit || n_probe < 0 || slots < 1 || n_parts < 1 || n_parts > 127 ||
        (!direct && (slots & (slots - 1))) || ((uintptr_t)probe_keys & 15) || (out_build_row && ((uintptr_t)out_build_row & 15)) ||
        ((uintptr_t)out_unit & 3) || (!build_payload != !out_payload) || (out_payload && ((uintptr_t)out_payload & 3))) {
        hs_set_error("hs_join_probe_unique: bad arguments (aligned buffers, 1..127 partitions)");
        return HS_E_ARG;
    }
    JoinProbeUniqueArgs A;
    A.t = JoinUniqueArgs{probe_keys, n_probe, const_cast<uint32_t*>(table), build_keys, slots, key_min, direct, nullptr};
    A.n_parts = n_parts;
    A.pad = 0;
    A.out_row = out_build_row;
    A.out_unit = out_unit;
    A.payload = build_payload;
    A.out_payload = out_payload;
    if (n_probe > 0)
        hipLaunchKernelGGL(k_join_probe_unique, dim3(hsj_grid((n_probe + 3) / 4, 256, 1 << 15)), dim3(256), 0, (hipStream_t)stream, A);
    HSJ_CHECK_LAUNCH("hs_join_probe_unique");
    return HS_OK;
}

// =====================================================================================================
// The byte table of the fused join (include/hipspark.h hs_join8): window partition + assembly in LDS
// =====================================================================================================
// Build rows arrive in any key order.  Workgroup g owns the contiguous row range [g * per, (g + 1) * per) in BOTH
// passes over the rows, so the per-(workgroup, window) counts of the histogram pass are exactly the segment sizes the
// scatter pass fills.  All counts are 32-bit (n_build < 2^32).
static constexpr int HSJ8_WSHIFT = 16;                       // log2(HS_JOIN8_WINDOW)
static constexpr int HSJ8_MAX_WINDOWS = 16384;               // the window histogram lives in LDS (64 KiB)
static_assert((1 << HSJ8_WSHIFT) == HS_JOIN8_WINDOW, "window size");

struct Join8Args {
    const int32_t* keys;
    const uint8_t* payload;     // optional
    int64_t n, seg_len;         // rows; rows per segment (valid: the first seg_counts[s] of a segment)
    const int64_t* seg_counts;  // optional (device)
    int64_t slots;
    int32_t key_min;
    int32_t n_win;
    int64_t per;                // rows per workgroup (a multiple of 4)
    int32_t staged, pad;        // the scatter stages a workgroup's tuples in LDS in window order (per <= HSJ8_STAGE_ROWS)
    uint32_t* hist;             // [n_groups][n_win]: tuples of workgroup g in window w
    uint32_t* offs;             // [n_groups][n_win]: where they start inside the window's segment (workgroup order)
    uint32_t* win_start;        // [n_win + 1]
    uint32_t* tuples;           // [n]: (offset in window) | payload << 24
    uint8_t* table;
    uint32_t* flags;
    const uint8_t* window_mask;  // optional [n_win]: 0 = this rank never probes the window - it is neither assembled nor written
};

// window of row i's key, or -1 (padding row of a segment / key outside the table: flagged)
// valid rows of the quad starting at row q (a multiple of 4; segments are multiples of 4 rows long, so a quad lies in one
// segment): one 32-bit division per QUAD (n_build < 2^32), not a 64-bit one per row
__device__ __forceinline__ int hsj8_quad_valid(const Join8Args& A, int64_t q) {
    if (!A.seg_counts) return 4;
    const uint32_t seg = (uint32_t)q / (uint32_t)A.seg_len;
    const int64_t left = A.seg_counts[seg] - (q - (int64_t)seg * A.seg_len);
    return left >= 4 ? 4 : (left < 0 ? 0 : (int)left);
}
__device__ __forceinline__ int hsj8_window(const Join8Args& A, bool valid, int32_t key, uint32_t& off, uint32_t& err) {
    if (!valid) return -1;
    const int64_t o = (int64_t)key - (int64_t)A.key_min;
    if ((uint64_t)o >= (uint64_t)A.slots) {
        err |= HS_FLAG_BAD_PROGRAM;
        return -1;
    }
    off = (uint32_t)o & (HS_JOIN8_WINDOW - 1);
    return (int)(o >> HSJ8_WSHIFT);
}

__global__ void __launch_bounds__(512) k_join8_hist(const Join8Args A) {
    extern __shared__ __align__(16) uint32_t s_hist[];
    for (int w = threadIdx.x; w < A.n_win; w += blockDim.x) s_hist[w] = 0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * A.per, r1 = r0 + A.per < A.n ? r0 + A.per : A.n;
    uint32_t err = 0;
    for (int64_t q = r0 + (int64_t)threadIdx.x * 4; q < r1; q += (int64_t)blockDim.x * 4) {
        const int4 kv = *reinterpret_cast<const int4*>(A.keys + q);  // buffers carry slack past the last row
        const int32_t k[4] = {kv.x, kv.y, kv.z, kv.w};
        const int nvalid = hsj8_quad_valid(A, q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (q + j >= r1) break;
            uint32_t off;
            const int w = hsj8_window(A, j < nvalid, k[j], off, err);
            if (w >= 0) atomicAdd(&s_hist[w], 1u);
        }
    }
    __syncthreads();
    uint32_t* mine = A.hist + (int64_t)blockIdx.x * A.n_win;
    for (int w = threadIdx.x; w < A.n_win; w += blockDim.x) mine[w] = s_hist[w];
    if (err) atomicOr(A.flags, err);
}

// per window: counts of the workgroups -> their offsets inside the window's segment (in workgroup order), total.
// A workgroup takes 16 consecutive windows; its 64 lane groups take 16 workgroup rows each (a row's load is 64
// contiguous bytes, all 16 of a lane in flight together), sums meet in LDS, a carry runs over chunks of 1024 rows.
// (First version: one lane per window walking all rows with a dependent store in between - 4 workgroups, latency-bound.)
__global__ void __launch_bounds__(1024) k_join8_scan_groups(const uint32_t* hist, uint32_t* offs, int32_t n_groups, int32_t n_win,
                                                            uint32_t* win_total) {
    constexpr int WPB = 16, SEGS = 1024 / WPB, RPS = 16;  // windows per workgroup, lane groups, rows per lane group and chunk
    __shared__ uint32_t s_sum[SEGS][WPB];
    __shared__ uint32_t s_carry[WPB];
    const int wl = threadIdx.x & (WPB - 1), seg = threadIdx.x / WPB;
    const int w = blockIdx.x * WPB + wl;
    const bool on = w < n_win;
    if (seg == 0) s_carry[wl] = 0;
    __syncthreads();
    for (int g0 = 0; g0 < n_groups; g0 += SEGS * RPS) {
        uint32_t c[RPS];
        uint32_t sum = 0;
        const int gs = g0 + seg * RPS;
#pragma unroll
        for (int k = 0; k < RPS; ++k) {
            const int g = gs + k;
            c[k] = (on && g < n_groups) ? hist[(int64_t)g * n_win + w] : 0u;
            sum += c[k];
        }
        s_sum[seg][wl] = sum;
        __syncthreads();
        uint32_t base = s_carry[wl];
        for (int k = 0; k < seg; ++k) base += s_sum[k][wl];
#pragma unroll
        for (int k = 0; k < RPS; ++k) {
            const int g = gs + k;
            if (on && g < n_groups) offs[(int64_t)g * n_win + w] = base;
            base += c[k];
        }
        __syncthreads();
        if (seg == SEGS - 1) s_carry[wl] = base;
        __syncthreads();
    }
    if (seg == 0 && on) win_total[w] = s_carry[wl];
}
// exclusive scan of the window totals in place (one workgroup; n_win <= 16384): win_start[n_win] = all rows
__global__ void __launch_bounds__(1024) k_join8_scan_windows(uint32_t* win_start, int32_t n_win) {
    __shared__ uint32_t s_part[16];
    __shared__ uint32_t s_base;
    const int tid = threadIdx.x, lane = tid & (HS_WAVE - 1), wv = tid / HS_WAVE;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int w0 = 0; w0 < n_win; w0 += 1024) {
        const int w = w0 + tid;
        const uint32_t c = w < n_win ? win_start[w] : 0;
        uint32_t x = c;
        for (int d = 1; d < HS_WAVE; d <<= 1) {
            const uint32_t t = __shfl_up(x, d, HS_WAVE);
            if (lane >= d) x += t;
        }
        if (lane == HS_WAVE - 1) s_part[wv] = x;
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (int k = 0; k < 16; ++k) {
            if (k < wv) before += s_part[k];
            all += s_part[k];
        }
        if (w < n_win) win_start[w] = s_base + before + x - c;
        __syncthreads();
        if (tid == 0) s_base += all;
        __syncthreads();
    }
    if (tid == 0) win_start[n_win] = s_base;
}

// Direct form: tuples go straight to their place (4-byte stores, a (workgroup, window) segment fills up piecemeal).
__global__ void __launch_bounds__(512) k_join8_scatter(const Join8Args A) {
    extern __shared__ __align__(16) uint32_t s_cur[];  // next free tuple position per window, for this workgroup
    const uint32_t* mine = A.offs + (int64_t)blockIdx.x * A.n_win;
    for (int w = threadIdx.x; w < A.n_win; w += blockDim.x) s_cur[w] = A.win_start[w] + mine[w];
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * A.per, r1 = r0 + A.per < A.n ? r0 + A.per : A.n;
    uint32_t err = 0;
    for (int64_t q = r0 + (int64_t)threadIdx.x * 4; q < r1; q += (int64_t)blockDim.x * 4) {
        const int4 kv = *reinterpret_cast<const int4*>(A.keys + q);
        const int32_t k[4] = {kv.x, kv.y, kv.z, kv.w};
        const int nvalid = hsj8_quad_valid(A, q);
        uint32_t pv = 0;
        if (A.payload) pv = *reinterpret_cast<const uint32_t*>(A.payload + q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (q + j >= r1) break;
            uint32_t off;
            const int w = hsj8_window(A, j < nvalid, k[j], off, err);
            if (w < 0) continue;
            const uint32_t code = (pv >> (8 * j)) & 0xffu;
            if (code == 0xffu) err |= HS_FLAG_BAD_PROGRAM;  // 0xff is the table's "no such key"
            const uint32_t pos = atomicAdd(&s_cur[w], 1u);
            A.tuples[pos] = off | (code << 24);
        }
    }
    if (err) atomicOr(A.flags, err);
}

// Staged form (a workgroup's rows fit LDS): the tuples are first laid out in LDS in window order - the workgroup's own
// counts give every window its run - and then each run leaves as ONE contiguous store of a wave, so a (workgroup,
// window) segment (64 bytes on average at sf=10) is written by one instruction instead of 16 stores spread over the
// workgroup's lifetime (partial sectors evicted from L2 in between were the build's write amplification).
static constexpr int HSJ8_STAGE_ROWS = 12288;   // 48 KiB of tuples
static constexpr int HSJ8_STAGE_WINDOWS = 4096; // + 2 x 16 KiB of run starts / cursors
__global__ void __launch_bounds__(512) k_join8_scatter_staged(const Join8Args A) {
    extern __shared__ __align__(16) uint32_t s_lds[];
    __shared__ uint32_t s_part[8];
    const int NW = A.n_win, tid = threadIdx.x, nthr = blockDim.x, lane = tid & (HS_WAVE - 1), wv = tid / HS_WAVE;
    uint32_t* s_base = s_lds;            // [NW + 1] start of every window's run in s_tup
    uint32_t* s_cur = s_lds + NW + 1;    // [NW] fill cursors, later the runs' global positions
    uint32_t* s_tup = s_cur + NW;        // [per]
    const uint32_t* cnt = A.hist + (int64_t)blockIdx.x * NW;
    for (int w = tid; w < NW; w += nthr) s_base[w] = cnt[w];
    __syncthreads();
    {   // exclusive scan of the counts in place: per-thread ranges, wave shuffles, one hop through LDS
        const int per_t = (NW + nthr - 1) / nthr;
        const int lo = tid * per_t < NW ? tid * per_t : NW, hi = lo + per_t < NW ? lo + per_t : NW;
        uint32_t sum = 0;
        for (int i = lo; i < hi; ++i) sum += s_base[i];
        uint32_t x = sum;
        for (int d = 1; d < HS_WAVE; d <<= 1) {
            const uint32_t t = __shfl_up(x, d, HS_WAVE);
            if (lane >= d) x += t;
        }
        if (lane == HS_WAVE - 1) s_part[wv] = x;
        __syncthreads();
        uint32_t run = x - sum;
        for (int k = 0; k < wv; ++k) run += s_part[k];
        for (int i = lo; i < hi; ++i) {
            const uint32_t c = s_base[i];
            s_base[i] = run;
            run += c;
        }
        if (tid == nthr - 1) s_base[NW] = run;  // the last thread's range ends the table (or is empty: run = total)
    }
    __syncthreads();
    for (int w = tid; w < NW; w += nthr) s_cur[w] = s_base[w];
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * A.per, r1 = r0 + A.per < A.n ? r0 + A.per : A.n;
    uint32_t err = 0;
    for (int64_t q = r0 + (int64_t)tid * 4; q < r1; q += (int64_t)nthr * 4) {
        const int4 kv = *reinterpret_cast<const int4*>(A.keys + q);
        const int32_t k[4] = {kv.x, kv.y, kv.z, kv.w};
        const int nvalid = hsj8_quad_valid(A, q);
        uint32_t pv = 0;
        if (A.payload) pv = *reinterpret_cast<const uint32_t*>(A.payload + q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (q + j >= r1) break;
            uint32_t off;
            const int w = hsj8_window(A, j < nvalid, k[j], off, err);
            if (w < 0) continue;
            const uint32_t code = (pv >> (8 * j)) & 0xffu;
            if (code == 0xffu) err |= HS_FLAG_BAD_PROGRAM;
            s_tup[atomicAdd(&s_cur[w], 1u)] = off | (code << 24);
        }
    }
    __syncthreads();
    const uint32_t* off_g = A.offs + (int64_t)blockIdx.x * NW;
    for (int w = tid; w < NW; w += nthr) s_cur[w] = A.win_start[w] + off_g[w];
    __syncthreads();
    // a quarter wave per window (a run holds ~13 tuples at sf=10): four runs leave per wave instruction
    for (int w = tid >> 4; w < NW; w += nthr >> 4) {
        const uint32_t b = s_base[w], e = s_base[w + 1];
        uint32_t* dst = A.tuples + s_cur[w];
        for (uint32_t i = b + (lane & 15); i < e; i += 16) dst[i - b] = s_tup[i];
    }
    if (err) atomicOr(A.flags, err);
}

// one workgroup per window: the window's slice of the table is assembled in LDS and leaves with 16-byte stores
__global__ void __launch_bounds__(512) k_join8_fill(const Join8Args A) {
    extern __shared__ __align__(16) uint32_t s_win[];  // HS_JOIN8_WINDOW bytes
    __shared__ uint32_t s_occupied;
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (A.window_mask && !A.window_mask[blockIdx.x]) {  // (workgroup-uniform) a window outside this rank's key stripes
        if (tid == 0 && A.win_start[blockIdx.x] != A.win_start[blockIdx.x + 1]) atomicOr(A.flags, HS_FLAG_BAD_PROGRAM);  // routed here by mistake
        return;
    }
    uint4* w4 = reinterpret_cast<uint4*>(s_win);
    for (int i = tid; i < HS_JOIN8_WINDOW / 16; i += nthr) w4[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    if (tid == 0) s_occupied = 0;
    __syncthreads();
    const uint32_t t0 = A.win_start[blockIdx.x], t1 = A.win_start[blockIdx.x + 1];
    uint8_t* bytes = reinterpret_cast<uint8_t*>(s_win);
    for (uint32_t t = t0 + tid; t < t1; t += nthr) {
        const uint32_t v = A.tuples[t];
        bytes[v & (HS_JOIN8_WINDOW - 1)] = (uint8_t)(v >> 24);
    }
    __syncthreads();
    uint4* out = reinterpret_cast<uint4*>(A.table + (int64_t)blockIdx.x * HS_JOIN8_WINDOW);
    uint32_t occ = 0;
    for (int i = tid; i < HS_JOIN8_WINDOW / 16; i += nthr) {
        const uint4 v = w4[i];
        out[i] = v;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // occupied bytes (!= 0xff) = non-zero bytes of ~x; exact SWAR: per byte ((b & 0x7f) + 0x7f) | b has its top
            // bit set iff b != 0, and no carry crosses a byte
            const uint32_t x = ~w[k];
            occ += __popc((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u);
        }
    }
    for (int d = HS_WAVE / 2; d >= 1; d >>= 1) occ += __shfl_down(occ, d, HS_WAVE);
    if ((tid & (HS_WAVE - 1)) == 0 && occ) atomicAdd(&s_occupied, occ);
    __syncthreads();
    // two tuples in one byte = one key twice on the build side
    if (tid == 0 && s_occupied != t1 - t0) atomicOr(A.flags, HS_FLAG_JOIN_DUP);
}

static int64_t hsj8_windows(int64_t slots) { return (slots + HS_JOIN8_WINDOW - 1) >> HSJ8_WSHIFT; }
// rows per workgroup of the two passes over the build rows; staged = the scatter's LDS form applies
static void hsj8_geometry(int64_t n, int64_t n_win, int64_t& per, int64_t& groups, bool& staged) {
    per = HSJ8_STAGE_ROWS;
    groups = n > 0 ? (n + per - 1) / per : 0;
    staged = n_win <= HSJ8_STAGE_WINDOWS && (size_t)groups * (size_t)n_win * 8 <= ((size_t)256 << 20);
    if (!staged) {  // big inputs: at most 1024 workgroups, tuples go straight to their place
        per = (n + 1023) / 1024;
        if (per < 8192) per = 8192;
        per = (per + 3) & ~(int64_t)3;
        groups = n > 0 ? (n + per - 1) / per : 0;
    }
}
extern "C" size_t hs_join8_table_bytes(int64_t slots) {
    return slots < 1 ? 0 : (size_t)hsj8_windows(slots) * HS_JOIN8_WINDOW;
}
extern "C" size_t hs_join8_ws_bytes(int64_t n_build, int64_t slots) {
    if (n_build < 0 || slots < 1) return 0;
    int64_t per, groups;
    bool staged;
    const int64_t n_win = hsj8_windows(slots);
    hsj8_geometry(n_build, n_win, per, groups, staged);
    return (size_t)(2 * groups * n_win + n_win + 1 + n_build) * 4 + 64;
}
static int join8_build_impl(void* stream, const int32_t* build_keys, const uint8_t* payload, int64_t n_build,
                            int64_t seg_len, const int64_t* seg_counts, int32_t key_min, int64_t slots, uint8_t* table,
                            void* ws, uint32_t* flags, const uint8_t* window_mask) {
    if (!build_keys || !table || !ws || !flags || n_build < 0 || n_build >= 0xffffffffll || slots < 1 ||
        slots > (1ll << 30) || ((uintptr_t)build_keys & 15) || ((uintptr_t)table & 15) || (payload && ((uintptr_t)payload & 3)) ||
        ((uintptr_t)ws & 3)) {
        hs_set_error("hs_join8_build: bad arguments (aligned buffers, n_build < 2^32, slots <= 2^30)");
        return HS_E_ARG;
    }
    if (seg_counts && (seg_len < 4 || (seg_len & 3) || n_build % seg_len)) {
        hs_set_error("hs_join8_build: segments must be equally long, a multiple of 4 rows");
        return HS_E_ARG;
    }
    const int64_t n_win = hsj8_windows(slots);
    if (n_win > HSJ8_MAX_WINDOWS) {
        hs_set_error("hs_join8_build: key range too wide");
        return HS_E_LIMIT;
    }
    int64_t per, groups;
    bool staged;
    hsj8_geometry(n_build, n_win, per, groups, staged);
    Join8Args A;
    A.keys = build_keys;
    A.payload = payload;
    A.n = n_build;
    A.seg_len = seg_counts ? seg_len : (n_build > 0 ? n_build : 1);
    A.seg_counts = seg_counts;
    A.slots = slots;
    A.key_min = key_min;
    A.n_win = (int32_t)n_win;
    A.per = per;
    A.staged = staged ? 1 : 0;
    A.pad = 0;
    A.hist = (uint32_t*)ws;
    A.offs = A.hist + groups * n_win;
    A.win_start = A.offs + groups * n_win;
    A.tuples = A.win_start + n_win + 1;
    A.table = table;
    A.flags = flags;
    A.window_mask = window_mask;
    hipStream_t s = (hipStream_t)stream;
    const size_t stage_lds = (size_t)(2 * n_win + 1 + HSJ8_STAGE_ROWS) * 4;
    static unsigned long long attr_set = 0;
    if (hs_first_on_device(attr_set)) {
        hipFuncSetAttribute((const void*)k_join8_fill, hipFuncAttributeMaxDynamicSharedMemorySize, HS_JOIN8_WINDOW);
        hipFuncSetAttribute((const void*)k_join8_hist, hipFuncAttributeMaxDynamicSharedMemorySize, HSJ8_MAX_WINDOWS * 4);
        hipFuncSetAttribute((const void*)k_join8_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, HSJ8_MAX_WINDOWS * 4);
        hipFuncSetAttribute((const void*)k_join8_scatter_staged, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (2 * HSJ8_STAGE_WINDOWS + 1 + HSJ8_STAGE_ROWS) * 4);
    }
    const size_t hist_lds = (size_t)n_win * 4;
    if (groups > 0) {
        hipLaunchKernelGGL(k_join8_hist, dim3((unsigned)groups), dim3(512), hist_lds, s, A);
        hipLaunchKernelGGL(k_join8_scan_groups, dim3((unsigned)((n_win + 15) / 16)), dim3(1024), 0, s, A.hist, A.offs,
                           (int32_t)groups, (int32_t)n_win, A.win_start);
    } else {
        hs_memset_async(A.win_start, 0, (size_t)(n_win + 1) * 4, s);
    }
    hipLaunchKernelGGL(k_join8_scan_windows, dim3(1), dim3(1024), 0, s, A.win_start, (int32_t)n_win);
    if (groups > 0) {
        if (staged) hipLaunchKernelGGL(k_join8_scatter_staged, dim3((unsigned)groups), dim3(512), stage_lds, s, A);
        else hipLaunchKernelGGL(k_join8_scatter, dim3((unsigned)groups), dim3(512), hist_lds, s, A);
    }
    hipLaunchKernelGGL(k_join8_fill, dim3((unsigned)n_win), dim3(512), HS_JOIN8_WINDOW, s, A);
    HSJ_CHECK_LAUNCH("hs_join8_build");
    return HS_OK;
}

extern "C" int hs_join8_build(void* stream, const int32_t* build_keys, const uint8_t* payload, int64_t n_build,
                              int64_t seg_len, const int64_t* seg_counts, int32_t key_min, int64_t slots, uint8_t* table,
                              void* ws, uint32_t* flags) {
    return join8_build_impl(stream, build_keys, payload, n_build, seg_len, seg_counts, key_min, slots, table, ws, flags, nullptr);
}
// The sharded form (N ranks): only the windows marked in window_mask[n_windows] (device) are assembled and written -
// the ones this rank's own probe rows can reach; the rest of the table's address range is never touched.
extern "C" int hs_join8_build_windows(void* stream, const int32_t* build_keys, const uint8_t* payload, int64_t n_build,
                                      int32_t key_min, int64_t slots, const uint8_t* window_mask, uint8_t* table, void* ws,
                                      uint32_t* flags) {
    if (!window_mask) {
        hs_set_error("hs_join8_build_windows: no window mask");
        return HS_E_ARG;
    }
    return join8_build_impl(stream, build_keys, payload, n_build, 0, nullptr, key_min, slots, table, ws, flags, window_mask);
}

// =====================================================================================================
// N ranks: the build side SHARDED by the probe side's key stripes (round 4)
// =====================================================================================================
// Reference: both join inputs are shuffled on the key (plan.py:186-189) and a JoinJob sees one partition of each
// (plan.py:99-109).  Here the probe rows never move; what decides where a BUILD row is needed is which rank's probe
// blocks can hold its key.  A probe table clustered on the key (TPC-H lineitem on l_orderkey) gives every block a key
// STRIPE [min, max]; block b lives on rank b % world.  A build row travels to the owner(s) of the stripe(s) containing
// its key - usually one, two when an order's lines straddle a block boundary, none when no probe block can match it.
//   hs_minmax_i32_units   the stripes: (min, max) of a key column per unit (file block)
//   hs_join8_route_count  per destination rank the number of (row, destination) pairs of this rank's build rows
//   hs_join8_route        the pairs themselves, grouped by destination: keys and payload codes ready for all_to_all_single
// Stripes arrive sorted by block id with non-decreasing min and max (the host checks that - otherwise the all-gather
// form runs): the first stripe whose max reaches the key is found by binary search, the following ones are walked
// while their min does not exceed it.

__global__ void __launch_bounds__(256) k_minmax_i32_units(const int32_t* v, const int64_t* unit_rows, int32_t* out) {
    __shared__ int32_t s_lo[4], s_hi[4];
    const int64_t b = unit_rows[blockIdx.x], e = unit_rows[blockIdx.x + 1];
    int32_t lo = 2147483647, hi = (-2147483647 - 1);
    for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        const int32_t x = v[i];
        lo = x < lo ? x : lo;
        hi = x > hi ? x : hi;
    }
    for (int d = HS_WAVE / 2; d >= 1; d >>= 1) {
        const int32_t l2 = __shfl_down(lo, d, HS_WAVE), h2 = __shfl_down(hi, d, HS_WAVE);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & (HS_WAVE - 1)) == 0) {
        s_lo[threadIdx.x / HS_WAVE] = lo;
        s_hi[threadIdx.x / HS_WAVE] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            lo = s_lo[w] < lo ? s_lo[w] : lo;
            hi = s_hi[w] > hi ? s_hi[w] : hi;
        }
        out[2 * blockIdx.x] = lo;       // a unit without rows: (INT32_MAX, INT32_MIN)
        out[2 * blockIdx.x + 1] = hi;
    }
}
extern "C" int hs_minmax_i32_units(void* stream, const int32_t* values, const int64_t* unit_rows_dev, int64_t n_units,
                                   int32_t* minmax) {
    if (!values || !unit_rows_dev || !minmax || n_units < 0 || n_units > (1 << 24)) {
        hs_set_error("hs_minmax_i32_units: bad arguments");
        return HS_E_ARG;
    }
    if (n_units > 0)
        hipLaunchKernelGGL(k_minmax_i32_units, dim3((unsigned)n_units), dim3(256), 0, (hipStream_t)stream, values, unit_rows_dev, minmax);
    HSJ_CHECK_LAUNCH("hs_minmax_i32_units");
    return HS_OK;
}

static constexpr int HSJ8_ROUTE_PER = 8192;      // build rows per workgroup of the two routing passes
static constexpr int HSJ8_ROUTE_STRIPES = 4096;  // stripes staged in LDS (3 x 16 KiB)
static constexpr int HSJ8_ROUTE_WORLD = 64;      // destinations are a 64-bit set

struct Join8RouteArgs {
    const int32_t* keys;
    const uint8_t* payload;  // optional
    int64_t n;
    const int32_t* stripe_min;
    const int32_t* stripe_max;
    const int32_t* stripe_owner;
    int32_t n_stripes, world;
    uint32_t* hist;        // [n_groups][world]
    uint32_t* offs;        // [n_groups][world]
    uint32_t* dest_start;  // [world + 1]
    const uint32_t* expect_start;  // optional [world + 1]: the sizes the host has agreed the collective on
    int32_t* out_keys;
    uint8_t* out_codes;
    int64_t out_cap;
    uint32_t* flags;
};

// the ranks that need build key k, as a bit set
__device__ __forceinline__ uint64_t hsj8_route_set(const int32_t* s_min, const int32_t* s_max, const int32_t* s_own, int n, int32_t k) {
    int lo = 0, hi = n;  // first stripe with max >= k
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (s_max[mid] < k) lo = mid + 1;
        else hi = mid;
    }
    uint64_t set = 0;
    for (int b = lo; b < n && s_min[b] <= k; ++b) set |= 1ull << s_own[b];
    return set;
}

template <bool SCATTER>
__global__ void __launch_bounds__(256) k_join8_route(const Join8RouteArgs A) {
    extern __shared__ __align__(16) int32_t s_stripes[];  // min[n] | max[n] | owner[n]
    __shared__ uint32_t s_cnt[HSJ8_ROUTE_WORLD];
    const int NS = A.n_stripes, tid = threadIdx.x, nthr = blockDim.x;
    int32_t *s_min = s_stripes, *s_max = s_stripes + NS, *s_own = s_stripes + 2 * NS;
    for (int i = tid; i < NS; i += nthr) {
        s_min[i] = A.stripe_min[i];
        s_max[i] = A.stripe_max[i];
        s_own[i] = A.stripe_owner[i];
    }
    if (tid < HSJ8_ROUTE_WORLD) {
        // scatter pass: the next free position of this workgroup's pairs inside every destination's share
        s_cnt[tid] = SCATTER && tid < A.world ? A.dest_start[tid] + A.offs[(int64_t)blockIdx.x * A.world + tid] : 0u;
    }
    __syncthreads();
    if (SCATTER && A.expect_start) {  // (uniform) the data no longer routes the way the agreed split sizes say: nothing is written
        bool stale = false;
        for (int d = 0; d <= A.world; ++d) stale = stale || A.dest_start[d] != A.expect_start[d];
        if (stale) {
            if (tid == 0 && blockIdx.x == 0) atomicOr(A.flags, HS_FLAG_ROUTE_STALE);
            return;
        }
    }
    const int64_t r0 = (int64_t)blockIdx.x * HSJ8_ROUTE_PER, r1 = r0 + HSJ8_ROUTE_PER < A.n ? r0 + HSJ8_ROUTE_PER : A.n;
    uint32_t err = 0;
    for (int64_t i = r0 + tid; i < r1; i += nthr) {
        const int32_t k = A.keys[i];
        uint64_t set = hsj8_route_set(s_min, s_max, s_own, NS, k);
        const uint8_t code = SCATTER && A.payload ? A.payload[i] : (uint8_t)0;
        while (set) {
            const int d = __ffsll((unsigned long long)set) - 1;
            set &= set - 1;
            const uint32_t pos = atomicAdd(&s_cnt[d], 1u);
            if constexpr (SCATTER) {
                if ((int64_t)pos < A.out_cap) {
                    A.out_keys[pos] = k;
                    if (A.out_codes) A.out_codes[pos] = code;
                } else {
                    err |= HS_FLAG_ROUTE_STALE;
                }
            }
        }
    }
    if constexpr (!SCATTER) {
        __syncthreads();
        if (tid < A.world) A.hist[(int64_t)blockIdx.x * A.world + tid] = s_cnt[tid];
    }
    if (err) atomicOr(A.flags, err);
}

extern "C" size_t hs_join8_route_ws_bytes(int64_t n_build, int32_t world) {
    if (n_build < 0 || world < 1 || world > HSJ8_ROUTE_WORLD) return 0;
    const int64_t groups = (n_build + HSJ8_ROUTE_PER - 1) / HSJ8_ROUTE_PER;
    return (size_t)(2 * groups * world) * 4 + 64;
}

static int join8_route_args(const char* who, const int32_t* keys, int64_t n, const int32_t* stripe_min, const int32_t* stripe_max,
                            const int32_t* stripe_owner, int32_t n_stripes, int32_t world, void* ws, uint32_t* dest_start,
                            Join8RouteArgs& A, int64_t& groups) {
    if ((!keys && n > 0) || !ws || !dest_start || ((uintptr_t)dest_start & 3) || n < 0 || n >= 0xffffffffll || world < 1 || world > HSJ8_ROUTE_WORLD || n_stripes < 0 ||
        n_stripes > HSJ8_ROUTE_STRIPES || (n_stripes > 0 && (!stripe_min || !stripe_max || !stripe_owner)) || ((uintptr_t)ws & 3)) {
        hs_set_error("%s: bad arguments (world <= %d, at most %d stripes)", who, HSJ8_ROUTE_WORLD, HSJ8_ROUTE_STRIPES);
        return HS_E_ARG;
    }
    groups = (n + HSJ8_ROUTE_PER - 1) / HSJ8_ROUTE_PER;
    A = Join8RouteArgs{};
    A.keys = keys;
    A.n = n;
    A.stripe_min = stripe_min;
    A.stripe_max = stripe_max;
    A.stripe_owner = stripe_owner;
    A.n_stripes = n_stripes;
    A.world = world;
    A.hist = (uint32_t*)ws;
    A.offs = A.hist + groups * world;
    A.dest_start = dest_start;
    return HS_OK;
}

// dest_start[world + 1] (device, caller-owned): where every destination's share of the routed pairs starts;
// dest_start[world] = all pairs.
extern "C" int hs_join8_route_count(void* stream, const int32_t* build_keys, int64_t n_build, const int32_t* stripe_min,
                                    const int32_t* stripe_max, const int32_t* stripe_owner, int32_t n_stripes, int32_t world,
                                    void* ws, uint32_t* dest_start) {
    Join8RouteArgs A;
    int64_t groups;
    const int rc = join8_route_args("hs_join8_route_count", build_keys, n_build, stripe_min, stripe_max, stripe_owner, n_stripes,
                                    world, ws, dest_start, A, groups);
    if (rc != HS_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (groups > 0) {
        hipLaunchKernelGGL(k_join8_route<false>, dim3((unsigned)groups), dim3(256), (size_t)n_stripes * 12, s, A);
        hipLaunchKernelGGL(k_join8_scan_groups, dim3((unsigned)((world + 15) / 16)), dim3(1024), 0, s, A.hist, A.offs, (int32_t)groups,
                           world, A.dest_start);
    } else {
        hs_memset_async(A.dest_start, 0, (size_t)(world + 1) * 4, s);
    }
    hipLaunchKernelGGL(k_join8_scan_windows, dim3(1), dim3(1024), 0, s, A.dest_start, world);
    HSJ_CHECK_LAUNCH("hs_join8_route_count");
    return HS_OK;
}

// after hs_join8_route_count on the same ws: the pairs, destination-major.  expect_start (optional, device): the split
// sizes the ranks agreed on in an earlier run - when the data routes differently now, HS_FLAG_ROUTE_STALE is raised and
// nothing is written (the collective that follows still has its agreed sizes; the query is then repeated from scratch).
extern "C" int hs_join8_route(void* stream, const int32_t* build_keys, const uint8_t* payload, int64_t n_build,
                              const int32_t* stripe_min, const int32_t* stripe_max, const int32_t* stripe_owner, int32_t n_stripes,
                              int32_t world, void* ws, const uint32_t* dest_start, const uint32_t* expect_start, int32_t* out_keys,
                              uint8_t* out_codes, int64_t out_cap, uint32_t* flags) {
    Join8RouteArgs A;
    int64_t groups;
    const int rc = join8_route_args("hs_join8_route", build_keys, n_build, stripe_min, stripe_max, stripe_owner, n_stripes, world,
                                    ws, const_cast<uint32_t*>(dest_start), A, groups);
    if (rc != HS_OK) return rc;
    if (!out_keys || !flags || out_cap < 0 || (payload && !out_codes)) {
        hs_set_error("hs_join8_route: no output buffers");
        return HS_E_ARG;
    }
    A.payload = payload;
    A.expect_start = expect_start;
    A.out_keys = out_keys;
    A.out_codes = out_codes;
    A.out_cap = out_cap;
    A.flags = flags;
    if (groups > 0)
        hipLaunchKernelGGL(k_join8_route<true>, dim3((unsigned)groups), dim3(256), (size_t)n_stripes * 12, (hipStream_t)stream, A);
    HSJ_CHECK_LAUNCH("hs_join8_route");
    return HS_OK;
}

__global__ void __launch_bounds__(256) k_remap_u8(const uint8_t* codes, int64_t n, const uint8_t* lut, uint8_t* out) {
    __shared__ uint8_t s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = s_lut[codes[i]];
}
extern "C" int hs_remap_u8(void* stream, const uint8_t* codes, int64_t n, const uint8_t* lut256, uint8_t* out) {
    if (!codes || !lut256 || !out || n < 0) {
        hs_set_error("hs_remap_u8: bad arguments");
        return HS_E_ARG;
    }
    if (n > 0) hipLaunchKernelGGL(k_remap_u8, dim3(hsj_grid(n, 256 * 8, 4096)), dim3(256), 0, (hipStream_t)stream, codes, n, lut256, out);
    HSJ_CHECK_LAUNCH("hs_remap_u8");
    return HS_OK;
}

// =====================================================================================================
// Synthetic orders table of BASELINE config 4 (bench / tests only; CPU twin: oracle/q45_oracle.c q4_gen_orders).
// Row j holds key(perm(j)), perm = an affine bijection of [0, n_total) (build order != key order), and a priority
// code in [0, 5).  key(o) = 32 * (o / 8) + o % 8 + 1: sparse like TPC-H order keys (8 used of every 32).
// =====================================================================================================
__device__ __forceinline__ uint64_t hsj_splitmix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
__global__ void __launch_bounds__(256) k_gen_orders(uint64_t seed, int64_t row0, int64_t n, uint64_t n_total, uint64_t mul,
                                                    uint64_t add, int32_t* okey, uint8_t* prio) {
    const uint64_t pseed = hsj_splitmix(seed + 0x7072696full);
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t j = (uint64_t)(row0 + k);
        const uint64_t o = (mul * j + add) % n_total;  // mul, j < 2^31: no overflow
        if (okey) okey[k] = (int32_t)(32 * (o / 8) + (o % 8) + 1);
        if (prio) prio[k] = (uint8_t)(hsj_splitmix(pseed + j) % 5);
    }
}
static uint64_t hsj_splitmix_host(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
extern "C" int hs_gen_orders(void* stream, uint64_t seed, int64_t row0, int64_t nrows, int64_t n_total, int32_t* orderkey,
                             uint8_t* priority_code) {
    if (nrows < 0 || row0 < 0 || n_total < 1 || n_total >= (1ll << 31) || row0 + nrows > n_total) {
        hs_set_error("hs_gen_orders: bad arguments (n_total < 2^31)");
        return HS_E_ARG;
    }
    uint64_t mul = 1;
    if (n_total > 1) {
        mul = hsj_splitmix_host(seed ^ 0x6f7264657273ull) % (uint64_t)n_total;
        if (mul < 2) mul = 2;
        for (;; ++mul) {
            uint64_t a = mul, b = (uint64_t)n_total;
            while (b) {
                const uint64_t t = a % b;
                a = b;
                b = t;
            }
            if (a == 1) break;
        }
        mul %= (uint64_t)n_total;
        if (!mul) mul = 1;
    }
    const uint64_t add = hsj_splitmix_host(seed + 77) % (uint64_t)n_total;
    if (nrows > 0)
        hipLaunchKernelGGL(k_gen_orders, dim3(hsj_grid(nrows, 256 * 4, 8192)), dim3(256), 0, (hipStream_t)stream, seed, row0, nrows,
                           (uint64_t)n_total, mul, add, orderkey, priority_code);
    HSJ_CHECK_LAUNCH("hs_gen_orders");
    return HS_OK;
}
